// C wrapper around the reference's own trajectory server: /root/reference/src/planner/traj_server/src/poly_traj_server.cpp is
// compiled WHOLE and UNMODIFIED, from where it lies, as part of this translation unit (its main() renamed by the preprocessor),
// against the stand-in ROS / message / Eigen headers of ref_shim/.  TEST INFRASTRUCTURE ONLY (oracle/_ref).
// What is the reference's: trajCallback's unpack loop (poly_traj_server.cpp:57-81: num_order + 1 coefficients per segment and axis
// from coef_x / coef_y / coef_z, durations from time[]), PolyTraj::addSegment / init / evaluate* (traj_utils/poly_traj.hpp), and
// cmdPubCallback (:23-55: t = max(0, odom stamp - trajectory stamp), position / velocity / acceleration into a PositionCommand).
// The test feeds it the message fields the packer (uavqp_pack_polynomial_trajectory) produced.
#define main ref_poly_traj_server_main
#include <traj_server/src/poly_traj_server.cpp>
#undef main

extern "C" {

// Delivers one quadrotor_msgs/PolynomialTrajectory to the reference's trajCallback.
void ref_traj_server_feed(unsigned trajectory_id, unsigned num_order, unsigned num_segment, const double* coef_x, const double* coef_y,
                          const double* coef_z, const double* time, int n_coef, double stamp) {
    auto msg = std::make_shared<quadrotor_msgs::PolynomialTrajectory>();
    msg->header.stamp = ros::Time(stamp);
    msg->trajectory_id = trajectory_id;
    msg->action = quadrotor_msgs::PolynomialTrajectory::ACTION_ADD;
    msg->num_order = num_order;
    msg->num_segment = num_segment;
    msg->coef_x.assign(coef_x, coef_x + n_coef);
    msg->coef_y.assign(coef_y, coef_y + n_coef);
    msg->coef_z.assign(coef_z, coef_z + n_coef);
    msg->time.assign(time, time + num_segment);
    quadrotor_msgs::PolynomialTrajectoryConstPtr cmsg = msg;
    trajCallback(cmsg);
}

// One tick of the reference's 100 Hz timer with an odometry message stamped odom_stamp: out[9] = position, velocity, acceleration of
// the PositionCommand it publishes.  Returns 1 if a command was published (has_trajectory_), else 0.
int ref_traj_server_tick(double odom_stamp, double* out9, int* ids3) {
    auto od = std::make_shared<nav_msgs::Odometry>();
    od->header.stamp = ros::Time(odom_stamp);
    nav_msgs::Odometry::ConstPtr cod = od;
    odomCallback(cod);
    const int before = ros::stub::publish_count<quadrotor_msgs::PositionCommand>();
    cmdPubCallback(ros::TimerEvent());
    if (ros::stub::publish_count<quadrotor_msgs::PositionCommand>() == before) return 0;
    const quadrotor_msgs::PositionCommand& c = ros::stub::last_published<quadrotor_msgs::PositionCommand>();
    out9[0] = c.position.x; out9[1] = c.position.y; out9[2] = c.position.z;
    out9[3] = c.velocity.x; out9[4] = c.velocity.y; out9[5] = c.velocity.z;
    out9[6] = c.acceleration.x; out9[7] = c.acceleration.y; out9[8] = c.acceleration.z;
    if (ids3) { ids3[0] = trajectory_id_; ids3[1] = num_order_; ids3[2] = num_segment_; }
    return 1;
}

double ref_traj_server_total_time(void) { return traj_.getTotalTIme(); }

}  // extern "C"
