// Test-only stand-in for the reference's <path_searching/rrt_star.h> (src/planner/path_searching, out of scope: SURVEY.md
// section 2): the members test_minimum_jerk.cpp:40-41,171,202-207 touches.  search() "finds" a fixed five-waypoint path
// between the start and the goal, so that the caller's optimiser section runs.  In the reference this header is also where
// the caller gets its ROS message types from.
#pragma once
#include <Eigen/Eigen>
#include <geometry_msgs/PoseStamped.h>
#include <memory>
#include <nav_msgs/Odometry.h>
#include <plan_env/grid_map.h>
#include <ros/ros.h>
#include <vector>
#include <visualization_msgs/Marker.h>
namespace path_searching {
class RRTStar {
  public:
    typedef std::shared_ptr<RRTStar> Ptr;
    void setParam(ros::NodeHandle&) {}
    void setGridMap(GridMap::Ptr&) {}
    void init() {}
    void reset() { optimal_path_.clear(); }
    int search(Eigen::Vector3d start, Eigen::Vector3d end, std::vector<Eigen::Vector3d>& path) {
        optimal_path_.clear();
        const double bend[5][3] = {{0, 0, 0}, {0.2, 0.5, 0.1}, {-0.1, 0.6, 0.2}, {0.3, 0.2, -0.1}, {0, 0, 0}};
        for (int i = 0; i < 5; ++i) {
            const double s = i / 4.0;
            optimal_path_.push_back(Eigen::Vector3d(start[0] + s * (end[0] - start[0]) + bend[i][0], start[1] + s * (end[1] - start[1]) + bend[i][1],
                                                    start[2] + s * (end[2] - start[2]) + bend[i][2]));
        }
        path = optimal_path_;
        return 1;
    }
    std::vector<Eigen::Vector3d> getOptimalPath() { return optimal_path_; }
  private:
    std::vector<Eigen::Vector3d> optimal_path_;
};
}  // namespace path_searching
