// C wrapper around the reference's own KinoAstar::isCollisionFree and KinoAstar::toPCL.  TEST INFRASTRUCTURE ONLY (oracle/_ref).
//
// What is the reference's here:
//   * the class declaration: /root/reference/src/planner/path_searching/include/path_searching/kino_astar.h, included from where it
//     lies (robot_r_, robot_h_, obs_, kdtree_ are ITS members; `private` is opened for this translation unit so that the wrapper can
//     fill them the way setParam / localCloudCallback do);
//   * the bodies of KinoAstar::isCollisionFree (kino_astar.cpp:721-758) and KinoAstar::toPCL (:761-774): oracle/Makefile cuts
//     exactly these two function definitions out of kino_astar.cpp AT BUILD TIME (awk, from the line of the signature to the closing
//     brace in column 0) into kino_astar_extract.inc in a TEMPORARY directory that exists for the duration of the compile only --
//     and this file includes it.  The rest of kino_astar.cpp (the search itself, ROS plumbing, grid map) is out of scope and would
//     need all of Eigen, PCL and ROS.
// What is NOT: Eigen (ref_shim/Eigen/Eigen: Vector3d / Matrix3d with Eigen's formulas for normalized() and the 3 x 3 inverse()),
// PCL / FLANN (ref_shim/rosmsgs/pcl/kdtree/kdtree_flann.h: exhaustive float radius search), ROS.
#define private public
#include <path_searching/kino_astar.h>
#undef private

#include <cmath>

namespace path_searching {
using std::cos;
using std::sin;
#include "kino_astar_extract.inc"
KinoAstar::~KinoAstar() {}   // the reference's destructor frees the search's node pool (kino_astar.cpp:776-782): nothing allocated here
}  // namespace path_searching

extern "C" {

// obs [n_obs][3] float64.  The reference receives its cloud as float32 PCL points and widens them (localCloudCallback,
// kino_astar.cpp:42-54): obs_ holds (double)(float)x, the kd-tree is built from toPCL(obs_).  as_float = 1 reproduces that;
// as_float = 0 keeps the doubles in obs_ (what the device entry points take) while the kd-tree still sees toPCL's floats.
void* ref_kino_create(const double* obs, int n_obs, double robot_r, double robot_h, int as_float) {
    auto* k = new path_searching::KinoAstar();
    k->allocated_node_num_ = 0;
    k->robot_r_ = robot_r;
    k->robot_h_ = robot_h;
    k->obs_.clear();
    for (int i = 0; i < n_obs; ++i) {
        Eigen::Vector3d pt;
        for (int c = 0; c < 3; ++c) pt(c) = as_float ? (double)(float)obs[3 * i + c] : obs[3 * i + c];
        k->obs_.push_back(pt);
    }
    path_searching::PCLPointCloud::Ptr cloud_ptr = boost::make_shared<path_searching::PCLPointCloud>(k->toPCL(k->obs_));   // :53
    k->kdtree_.setInputCloud(cloud_ptr);                                                                                    // :54
    return k;
}

void ref_kino_destroy(void* h) { delete static_cast<path_searching::KinoAstar*>(h); }

// KinoAstar::isCollisionFree(pt, acc): 1 = free, 0 = collides
int ref_kino_is_collision_free(void* h, const double* pt, const double* acc) {
    auto* k = static_cast<path_searching::KinoAstar*>(h);
    return k->isCollisionFree(Eigen::Vector3d(pt[0], pt[1], pt[2]), Eigen::Vector3d(acc[0], acc[1], acc[2])) ? 1 : 0;
}

void ref_kino_is_collision_free_batch(void* h, int n, const double* pts, const double* accs, int* out) {
    for (int i = 0; i < n; ++i) out[i] = ref_kino_is_collision_free(h, pts + 3 * i, accs + 3 * i);
}

}  // extern "C"
