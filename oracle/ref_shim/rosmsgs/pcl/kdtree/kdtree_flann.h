// Stand-in for <pcl/kdtree/kdtree_flann.h> -- TEST INFRASTRUCTURE ONLY (PCL and FLANN are absent from this image).
// pcl::PointXYZ (float x, y, z), pcl::PointCloud<PointT> (width, height, points, Ptr, makeShared) and a KdTreeFLANN whose
// radiusSearch is an EXHAUSTIVE scan with the arithmetic of the real one, as far as it is known without the sources:
//   pcl::KdTreeFLANN<PointT>::radiusSearch(point, double radius, k_indices, k_sqr_distances, max_nn = 0) hands
//   static_cast<float>(radius * radius) to flann::Index<L2_Simple<float>>::radiusSearch; L2_Simple accumulates (a - b)^2 over
//   x, y, z in FLOAT; flann's RadiusResultSet keeps a point when dist < radius (strict; from memory of flann 1.8
//   result_set.h -- a point exactly on the float sphere may differ, which cannot change KinoAstar::isCollisionFree's verdict:
//   the ellipsoid lies strictly inside the search sphere); results are returned sorted by distance.
#pragma once
#include <algorithm>
#include <memory>
#include <utility>
#include <vector>
namespace pcl {
struct PointXYZ { float x = 0.f, y = 0.f, z = 0.f; };
template <typename PointT>
struct PointCloud {
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    unsigned int width = 0, height = 0;
    std::vector<PointT> points;
    Ptr makeShared() const { return std::make_shared<PointCloud<PointT>>(*this); }
};
template <typename PointT>
class KdTreeFLANN {
  public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& cloud) { cloud_ = cloud; }
    int radiusSearch(const PointT& p, double radius, std::vector<int>& k_indices, std::vector<float>& k_sqr_distances,
                     unsigned int max_nn = 0) const {
        k_indices.clear();
        k_sqr_distances.clear();
        if (!cloud_) return 0;
        const float r2 = static_cast<float>(radius * radius);
        std::vector<std::pair<float, int>> found;
        for (size_t i = 0; i < cloud_->points.size(); ++i) {
            const PointT& q = cloud_->points[i];
            float d = 0.f, t;
            t = p.x - q.x; d += t * t;
            t = p.y - q.y; d += t * t;
            t = p.z - q.z; d += t * t;
            if (d < r2) found.emplace_back(d, static_cast<int>(i));
        }
        std::sort(found.begin(), found.end());
        if (max_nn > 0 && found.size() > max_nn) found.resize(max_nn);
        for (const auto& f : found) { k_indices.push_back(f.second); k_sqr_distances.push_back(f.first); }
        return static_cast<int>(k_indices.size());
    }
  private:
    typename PointCloud<PointT>::ConstPtr cloud_;
};
}  // namespace pcl
