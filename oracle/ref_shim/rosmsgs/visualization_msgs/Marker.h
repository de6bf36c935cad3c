// Stand-in -- TEST INFRASTRUCTURE ONLY: the Marker member of path_searching/kino_astar.h is only declared, never used by oracle/_ref.
#pragma once
namespace visualization_msgs { struct Marker {}; }
