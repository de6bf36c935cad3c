// Stand-in -- TEST INFRASTRUCTURE ONLY (boost is absent): kino_astar.cpp:53 wraps the cloud with boost::make_shared.
#pragma once
#include <memory>
namespace boost { using std::make_shared; using std::shared_ptr; }
