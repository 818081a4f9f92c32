/*
 * lmcma_ros.h — the step after the path: the converged path and the map as the ROS
 * messages the reference's node publishes, filled into plain C structures that carry
 * the fields of those messages one to one.  Nothing here needs ROS: a node copies the
 * fields into visualization_msgs::Marker / sensor_msgs::PointCloud2 /
 * octomap_msgs::Octomap and publishes them (ros/lmcma_planner_node.cpp does exactly
 * that and is compiled only where catkin is found).
 *
 * S/ = sample_based_optimisation_based_path_planner/src/
 *      sample_based_optimisation_based_path_planner.cpp of the reference.
 * Topic names, frame and marker bookkeeping are the reference's:
 *   "visualization_marker"  visualization_msgs::Marker, frame "base_link",
 *                           ns "my_namespace", id 1, action ADD        S/:1470-1482
 *   "pointcloud2"           sensor_msgs::PointCloud2 (latched), frame "base_link",
 *                           the layout pcl::toROSMsg gives pcl::PointXYZ   S/:1060-1072
 *   "octomapbinary"         octomap_msgs::Octomap (latched), frame "base_link",
 *                           octomap_msgs::binaryMapToMsg of an OcTree      S/:1076-1090
 * The reference draws its markers as CUBE / CUBE_LIST (S/:1289, 1480); a path is a
 * LINE_STRIP through start, the waypoints and goal.
 *
 * All functions are host-only, return 0 on success and a negative code on failure
 * (-1 bad argument, -4 capacity too small).  The octree function uses temporary host memory
 * of about 1.15 bytes per cell; the others allocate nothing.
 */
#ifndef LMCMA_ROS_H
#define LMCMA_ROS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMCMA_ROS_TOPIC_MARKER "visualization_marker"   /* S/:1472 */
#define LMCMA_ROS_TOPIC_POINTCLOUD2 "pointcloud2"       /* S/:1063 */
#define LMCMA_ROS_TOPIC_OCTOMAP "octomapbinary"         /* S/:1079 */
#define LMCMA_ROS_FRAME "base_link"                     /* S/:1067, 1080, 1470 */
#define LMCMA_ROS_MARKER_NS "my_namespace"              /* S/:1478 */

/* visualization_msgs/Marker constants */
#define LMCMA_ROS_MARKER_LINE_STRIP 4
#define LMCMA_ROS_MARKER_CUBE_LIST 6
#define LMCMA_ROS_MARKER_ADD 0
#define LMCMA_ROS_MARKER_DELETE 2
/* sensor_msgs/PointField datatype */
#define LMCMA_ROS_FLOAT32 7

/* visualization_msgs/Marker, the fields a LINE_STRIP uses.  points: n_points triples
 * (x, y, z) of geometry_msgs/Point in the caller's buffer.                            */
typedef struct {
    char frame_id[32];            /* header.frame_id */
    char ns[32];
    int32_t id;
    int32_t type;                 /* LMCMA_ROS_MARKER_LINE_STRIP */
    int32_t action;               /* LMCMA_ROS_MARKER_ADD */
    double position[3];           /* pose.position */
    double orientation[4];        /* pose.orientation x, y, z, w = (0, 0, 0, 1) as S/:1284-1287 */
    double scale[3];              /* scale.x = line width; y, z unused by LINE_STRIP */
    float color[4];               /* r, g, b, a */
    int32_t n_points;
    double *points;               /* [n_points][3] */
} lmcma_ros_marker;

/* sensor_msgs/PointCloud2 with the three FLOAT32 fields x, y, z */
typedef struct {
    char frame_id[32];
    uint32_t height, width;       /* 1, number of points */
    char field_name[3][4];        /* "x", "y", "z" */
    uint32_t field_offset[3];     /* 0, 4, 8 */
    uint8_t field_datatype[3];    /* LMCMA_ROS_FLOAT32 */
    uint32_t field_count[3];      /* 1 */
    uint8_t is_bigendian;         /* 0 */
    uint32_t point_step;          /* 16: pcl::PointXYZ is padded to four floats */
    uint32_t row_step;            /* point_step * width */
    uint8_t is_dense;             /* 1 */
    size_t data_bytes;            /* row_step * height */
    uint8_t *data;
} lmcma_ros_pointcloud2;

/* octomap_msgs/Octomap */
typedef struct {
    char frame_id[32];
    uint8_t binary;               /* 1 */
    char id[16];                  /* "OcTree" */
    double resolution;
    size_t data_bytes;
    int8_t *data;
} lmcma_ros_octomap;

/* The path of a parameter vector as a LINE_STRIP: start, the W waypoints, goal, in map
 * coordinates times cell_size (metres per cell).  x: dimension-major [dim0: wp0..wpW-1 |
 * dim1: ...] like every parameter vector of the optimiser (lmcma.cpp:786-791), dims in
 * 1..3 (missing coordinates are 0).  points: room for (W + 2) triples.  line_width:
 * scale.x in metres.                                                                  */
int lmcma_ros_path_marker(const double *x, int dims, int waypoints, const double *start, const double *goal,
                          double cell_size, double line_width, double *points, int points_capacity,
                          lmcma_ros_marker *out);
/* the same marker with action DELETE and no points (wipe_marker, S/:1049-1057) */
int lmcma_ros_path_marker_delete(lmcma_ros_marker *out);

/* Occupied cells of occ [nz][ny][nx] (non-zero = occupied) as one point each, at the
 * cell centre (x + 0.5, y + 0.5, z + 0.5) * cell_size + origin.  With data = NULL only
 * the sizes are filled in (width, row_step, data_bytes): call twice.                  */
int lmcma_ros_grid_pointcloud2(const uint8_t *occ, int nx, int ny, int nz, double cell_size, const double *origin,
                               uint8_t *data, size_t data_capacity, lmcma_ros_pointcloud2 *out);

/* occ as the binary stream of an octomap::OcTree of that resolution whose leaves are the
 * cells: occupied cells occupied, free cells free, everything outside the grid unknown,
 * eight equal leaves merged into their parent as OcTree::prune does.  Cell (x, y, z) is
 * the voxel with key (x, y, z) + key_origin + 32768, i.e. the metric cube
 * [(x + key_origin[0]) res, (x + key_origin[0] + 1) res) ... (key_origin NULL: zeros);
 * the grid must fit the 16-level tree.  The stream is what OcTree::writeBinaryData
 * emits, which binaryMapToMsg puts into the message: per inner node two bytes, two bits
 * per child (bit 2i, bit 2i+1) = (0,0) unknown, (1,0) free leaf, (0,1) occupied leaf,
 * (1,1) inner node; child i covers the upper half in x / y / z when bit 0 / 1 / 2 of i
 * is set; the children that are inner nodes follow depth first.  octomap is not part of
 * the reference's sources: the format is restated from its published description and no
 * reference output pins it.  With data = NULL only data_bytes is filled in.           */
int lmcma_ros_grid_octomap_binary(const uint8_t *occ, int nx, int ny, int nz, double resolution,
                                  const int *key_origin, int8_t *data, size_t data_capacity,
                                  lmcma_ros_octomap *out);

#ifdef __cplusplus
}
#endif
#endif
