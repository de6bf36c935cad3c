// Stand-in -- TEST INFRASTRUCTURE ONLY: path_searching/kino_astar.h declares a GridMap::Ptr member (plan_env is out of scope).
// The real header pulls sensor_msgs in through pcl_conversions (plan_env/grid_map.h:18), which kino_astar.h relies on.
#pragma once
#include <memory>
#include <sensor_msgs/PointCloud2.h>
class GridMap { public: typedef std::shared_ptr<GridMap> Ptr; };
