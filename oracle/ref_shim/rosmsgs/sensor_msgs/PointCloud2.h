// Stand-in -- TEST INFRASTRUCTURE ONLY: only named in a declaration of path_searching/kino_astar.h (localCloudCallback).
#pragma once
#include <memory>
namespace sensor_msgs {
struct PointCloud2 { typedef std::shared_ptr<const PointCloud2> ConstPtr; };
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
}
