// ROS node around the GPU planner: plans a path on a map with the LM-CMA optimiser of
// include/lmcma.hpp and publishes what the reference's node publishes, on its topics and in
// its frame (sample_based_optimisation_based_path_planner.cpp, cited as S/):
//   visualization_marker  visualization_msgs/Marker   the converged path, LINE_STRIP  (S/:1470-1482, 1536)
//   pointcloud2           sensor_msgs/PointCloud2     the occupied cells, latched     (S/:1060-1072)
//   octomapbinary         octomap_msgs/Octomap        the map as a binary OcTree      (S/:1076-1090)
// The messages are filled by the host functions of include/lmcma_ros.h into plain structures
// and copied field by field here; this file is the only one that needs ROS and is compiled
// only where catkin is found (ros/CMakeLists.txt).
//
// Parameters (private namespace): ~map (path of a .bmp or .binvox file; empty: the two-bar
// map of S/:269-297), ~size (cells per side of the two-bar map, default 100), ~waypoints (32),
// ~lambda (256), ~vectors (16), ~generations (500), ~cell_size (metres per cell, 0.01: the
// resolution of the reference's OcTree, S/:1135), ~device (0).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <geometry_msgs/Point.h>
#include <octomap_msgs/Octomap.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/PointField.h>
#include <visualization_msgs/Marker.h>

#include "lmcma.hpp"
#include "lmcma_opt.h"
#include "lmcma_ros.h"

namespace {

bool ends_with(const std::string &s, const char *suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

visualization_msgs::Marker to_msg(const lmcma_ros_marker &m)
{
    visualization_msgs::Marker msg;
    msg.header.frame_id = m.frame_id;
    msg.header.stamp = ros::Time();          // S/:1477
    msg.ns = m.ns;
    msg.id = m.id;
    msg.type = m.type;
    msg.action = m.action;
    msg.pose.position.x = m.position[0];
    msg.pose.position.y = m.position[1];
    msg.pose.position.z = m.position[2];
    msg.pose.orientation.x = m.orientation[0];
    msg.pose.orientation.y = m.orientation[1];
    msg.pose.orientation.z = m.orientation[2];
    msg.pose.orientation.w = m.orientation[3];
    msg.scale.x = m.scale[0];
    msg.scale.y = m.scale[1];
    msg.scale.z = m.scale[2];
    msg.color.r = m.color[0];
    msg.color.g = m.color[1];
    msg.color.b = m.color[2];
    msg.color.a = m.color[3];
    msg.lifetime = ros::Duration();          // S/:1268
    msg.points.resize(m.n_points);
    for (int k = 0; k < m.n_points; ++k) {
        msg.points[k].x = m.points[3 * k];
        msg.points[k].y = m.points[3 * k + 1];
        msg.points[k].z = m.points[3 * k + 2];
    }
    return msg;
}

sensor_msgs::PointCloud2 to_msg(const lmcma_ros_pointcloud2 &c)
{
    sensor_msgs::PointCloud2 msg;
    msg.header.frame_id = c.frame_id;
    msg.header.stamp = ros::Time::now();
    msg.height = c.height;
    msg.width = c.width;
    msg.fields.resize(3);
    for (int f = 0; f < 3; ++f) {
        msg.fields[f].name = c.field_name[f];
        msg.fields[f].offset = c.field_offset[f];
        msg.fields[f].datatype = c.field_datatype[f];
        msg.fields[f].count = c.field_count[f];
    }
    msg.is_bigendian = c.is_bigendian;
    msg.point_step = c.point_step;
    msg.row_step = c.row_step;
    msg.is_dense = c.is_dense;
    msg.data.assign(c.data, c.data + c.data_bytes);
    return msg;
}

octomap_msgs::Octomap to_msg(const lmcma_ros_octomap &o)
{
    octomap_msgs::Octomap msg;
    msg.header.frame_id = o.frame_id;
    msg.header.stamp = ros::Time::now();     // S/:1083
    msg.binary = o.binary;
    msg.id = o.id;
    msg.resolution = o.resolution;
    msg.data.assign(o.data, o.data + o.data_bytes);
    return msg;
}

}  // namespace

int main(int argc, char **argv)
{
    ros::init(argc, argv, "lmcma_planner");
    ros::NodeHandle node_handle;
    ros::NodeHandle pnh("~");

    std::string map_path;
    int size = 100, W = 32, lambda = 256, m = 16, generations = 500, device = 0;
    double cell_size = 0.01;
    pnh.param<std::string>("map", map_path, "");
    pnh.param("size", size, size);
    pnh.param("waypoints", W, W);
    pnh.param("lambda", lambda, lambda);
    pnh.param("vectors", m, m);
    pnh.param("generations", generations, generations);
    pnh.param("cell_size", cell_size, cell_size);
    pnh.param("device", device, device);

    // ---- the map: [z][y][x] bytes, non-zero = occupied
    int nx = size, ny = size, nz = 1;
    std::vector<uint8_t> occ;
    if (map_path.empty()) {
        occ.assign((size_t)size * size, 0);
        lmcma_grid_two_bar(size, 1, occ.data());
    } else if (ends_with(map_path, ".binvox")) {
        int dims[3];
        double translate[3], scale;
        if (lmcma_grid_read_binvox(map_path.c_str(), dims, translate, &scale, nullptr, 0) < 0) {
            ROS_ERROR("cannot read %s", map_path.c_str());
            return 1;
        }
        nx = dims[0]; ny = dims[1]; nz = dims[2];
        occ.assign((size_t)nx * ny * nz, 0);
        if (lmcma_grid_read_binvox(map_path.c_str(), dims, translate, &scale, occ.data(), (long)occ.size()) < 0) {
            ROS_ERROR("cannot read %s", map_path.c_str());
            return 1;
        }
    } else {
        if (lmcma_grid_read_bmp(map_path.c_str(), &nx, &ny, nullptr, 0) < 0) {
            ROS_ERROR("cannot read %s", map_path.c_str());
            return 1;
        }
        occ.assign((size_t)nx * ny, 0);
        if (lmcma_grid_read_bmp(map_path.c_str(), &nx, &ny, occ.data(), (long)occ.size()) < 0) {
            ROS_ERROR("cannot read %s", map_path.c_str());
            return 1;
        }
    }
    const int D = nz > 1 ? 3 : 2;
    const int N = D * W;

    // ---- the planning problem of S/:697-706 scaled to the map: (nx-1, 0[, 0]) -> (0, ny-1[, nz-1])
    LMCMATrajectoryProblem problem;
    problem.dims = D;
    problem.waypoints = W;
    problem.nx = nx; problem.ny = ny; problem.nz = nz;
    problem.cells = occ.data();
    problem.start[0] = nx - 1.0; problem.start[1] = 0.0;      problem.start[2] = 0.0;
    problem.goal[0] = 0.0;       problem.goal[1] = ny - 1.0;  problem.goal[2] = nz > 1 ? nz - 1.0 : 0.0;
    const double extent[3] = {nx - 1.0, ny - 1.0, nz - 1.0};
    std::vector<double> x0(N), lo(N, 0.0), hi(N);
    for (int d = 0; d < D; ++d)
        for (int k = 0; k < W; ++k) {
            const double t = (k + 1.0) / (W + 1.0);
            x0[(size_t)d * W + k] = problem.start[d] + t * (problem.goal[d] - problem.start[d]);
            hi[(size_t)d * W + k] = extent[d];
        }

    LMCMA opt(x0.data(), lambda, lo.data(), hi.data(), nx / 20.0, 0, 42);
    opt.setDirectionVectors(m);
    opt.setDevice(device);
    opt.usePhiloxSampling(true);
    opt.useDeviceUpdate(true);
    opt.setTrajectoryProblem(problem);
    opt.init(N);
    opt.step(generations);   // returns early once the stop rule fires (lmcma.cpp:426-429)
    std::vector<double> path(N);
    opt.mean(path.data());
    ROS_INFO("planned: %d generations, %d evaluations, best cost %.4f", opt.generation(), opt.counteval, opt.BestF);

    // ---- the messages
    std::vector<double> points(3 * (size_t)(W + 2));
    lmcma_ros_marker marker;
    if (lmcma_ros_path_marker(path.data(), D, W, problem.start, problem.goal, cell_size, 2.0 * cell_size,
                              points.data(), W + 2, &marker) != 0) {
        ROS_ERROR("lmcma_ros_path_marker failed");
        return 1;
    }
    lmcma_ros_pointcloud2 cloud;
    lmcma_ros_grid_pointcloud2(occ.data(), nx, ny, nz, cell_size, nullptr, nullptr, 0, &cloud);
    std::vector<uint8_t> cloud_data(cloud.data_bytes + 1);
    lmcma_ros_grid_pointcloud2(occ.data(), nx, ny, nz, cell_size, nullptr, cloud_data.data(), cloud_data.size(), &cloud);
    lmcma_ros_octomap tree;
    std::vector<int8_t> tree_data;
    const bool have_tree =
        lmcma_ros_grid_octomap_binary(occ.data(), nx, ny, nz, cell_size, nullptr, nullptr, 0, &tree) == 0;
    if (have_tree) {
        tree_data.resize(tree.data_bytes + 1);
        lmcma_ros_grid_octomap_binary(occ.data(), nx, ny, nz, cell_size, nullptr, tree_data.data(), tree_data.size(),
                                      &tree);
    }

    ros::Publisher vis_pub = node_handle.advertise<visualization_msgs::Marker>(LMCMA_ROS_TOPIC_MARKER, 0, true);
    ros::Publisher cloud_pub = node_handle.advertise<sensor_msgs::PointCloud2>(LMCMA_ROS_TOPIC_POINTCLOUD2, 1, true);
    ros::Publisher tree_pub = node_handle.advertise<octomap_msgs::Octomap>(LMCMA_ROS_TOPIC_OCTOMAP, 1, true);
    cloud_pub.publish(to_msg(cloud));
    if (have_tree) tree_pub.publish(to_msg(tree));
    vis_pub.publish(to_msg(marker));
    ros::spin();   // the topics are latched: late subscribers still get the messages
    return 0;
}
