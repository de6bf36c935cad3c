// Test-only stand-in for the reference's <plan_env/grid_map.h> (src/planner/plan_env, out of scope: SURVEY.md section 2):
// the two members test_minimum_jerk.cpp:199-200 touches.
#pragma once
#include <memory>
#include <ros/ros.h>
class GridMap {
  public:
    typedef std::shared_ptr<GridMap> Ptr;
    void initMap(ros::NodeHandle&) {}
};
