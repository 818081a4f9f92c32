// Host side of include/lmcma_ros.h: the converged path and the occupancy map laid out as
// the fields of the ROS messages the reference's node publishes.  Nothing here touches
// the GPU or needs ROS.
//
// S/ = sample_based_optimisation_based_path_planner/src/
//      sample_based_optimisation_based_path_planner.cpp of the reference.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "lmcma_ros.h"

namespace {

void put(char *dst, size_t cap, const char *src)
{
    snprintf(dst, cap, "%s", src);
}

// One level of the octree over the grid, in absolute key space: node (ix, iy, iz) of level l
// covers keys [i << l, (i + 1) << l) per axis.  Only the nodes that meet the grid are stored.
enum : uint8_t { kUnknown = 0, kFree = 1, kOccupied = 2, kInner = 3 };

struct Level {
    int lo[3], cnt[3];
    std::vector<uint8_t> status;
    uint8_t at(int ix, int iy, int iz) const
    {
        const int x = ix - lo[0], y = iy - lo[1], z = iz - lo[2];
        if (x < 0 || y < 0 || z < 0 || x >= cnt[0] || y >= cnt[1] || z >= cnt[2]) return kUnknown;
        return status[((size_t)z * cnt[1] + y) * cnt[0] + x];
    }
};

constexpr int kTreeDepth = 16;          // octomap::OcTree: 16 levels below the root
constexpr int kTreeMaxVal = 32768;      // key of the metric origin

// the two bytes of an inner node and its inner children after it, depth first
// (OcTree::writeBinaryNode)
void emit(const std::vector<Level> &lv, int l, int ix, int iy, int iz, int8_t *data, size_t cap, size_t &n)
{
    unsigned bits = 0;
    for (int i = 0; i < 8; ++i) {
        const uint8_t s = lv[l - 1].at(2 * ix + (i & 1), 2 * iy + ((i >> 1) & 1), 2 * iz + ((i >> 2) & 1));
        // (bit 2i, bit 2i + 1): free (1, 0), occupied (0, 1), inner (1, 1)
        const unsigned two = s == kFree ? 1u : s == kOccupied ? 2u : s == kInner ? 3u : 0u;
        bits |= two << (2 * i);
    }
    if (data && n + 2 <= cap) {
        data[n] = (int8_t)(bits & 0xff);           // children 0..3
        data[n + 1] = (int8_t)((bits >> 8) & 0xff);  // children 4..7
    }
    n += 2;
    if (l == 1) return;
    for (int i = 0; i < 8; ++i) {
        const int cx = 2 * ix + (i & 1), cy = 2 * iy + ((i >> 1) & 1), cz = 2 * iz + ((i >> 2) & 1);
        if (lv[l - 1].at(cx, cy, cz) == kInner) emit(lv, l - 1, cx, cy, cz, data, cap, n);
    }
}

}  // namespace

extern "C" {

int lmcma_ros_path_marker(const double *x, int dims, int waypoints, const double *start, const double *goal,
                          double cell_size, double line_width, double *points, int points_capacity,
                          lmcma_ros_marker *out)
{
    if (!x || !start || !goal || !points || !out || dims < 1 || dims > 3 || waypoints < 0 || !(cell_size > 0.0))
        return -1;
    const int n = waypoints + 2;
    if (points_capacity < n) return -4;
    memset(out, 0, sizeof(*out));
    put(out->frame_id, sizeof(out->frame_id), LMCMA_ROS_FRAME);
    put(out->ns, sizeof(out->ns), LMCMA_ROS_MARKER_NS);
    out->id = 1;                                  // S/:1479
    out->type = LMCMA_ROS_MARKER_LINE_STRIP;
    out->action = LMCMA_ROS_MARKER_ADD;           // S/:1482
    out->orientation[3] = 1.0;                    // S/:1284-1287
    out->scale[0] = line_width;
    out->color[0] = 1.f;                          // opaque red
    out->color[3] = 1.f;                          // S/:1266
    for (int k = 0; k < n; ++k) {
        for (int d = 0; d < 3; ++d) {
            double v = 0.0;
            if (d < dims) v = k == 0 ? start[d] : k == n - 1 ? goal[d] : x[(size_t)d * waypoints + (k - 1)];
            points[3 * k + d] = v * cell_size;
        }
    }
    out->n_points = n;
    out->points = points;
    return 0;
}

int lmcma_ros_path_marker_delete(lmcma_ros_marker *out)
{
    if (!out) return -1;
    memset(out, 0, sizeof(*out));
    put(out->frame_id, sizeof(out->frame_id), LMCMA_ROS_FRAME);
    put(out->ns, sizeof(out->ns), LMCMA_ROS_MARKER_NS);
    out->id = 1;
    out->type = LMCMA_ROS_MARKER_LINE_STRIP;
    out->action = LMCMA_ROS_MARKER_DELETE;        // S/:1053
    out->orientation[3] = 1.0;
    return 0;
}

int lmcma_ros_grid_pointcloud2(const uint8_t *occ, int nx, int ny, int nz, double cell_size, const double *origin,
                               uint8_t *data, size_t data_capacity, lmcma_ros_pointcloud2 *out)
{
    if (!occ || !out || nx < 1 || ny < 1 || nz < 1 || !(cell_size > 0.0)) return -1;
    const size_t cells = (size_t)nx * ny * nz;
    size_t n = 0;
    for (size_t c = 0; c < cells; ++c) n += occ[c] != 0;
    memset(out, 0, sizeof(*out));
    put(out->frame_id, sizeof(out->frame_id), LMCMA_ROS_FRAME);   // S/:1067-1068
    out->height = 1;
    out->width = (uint32_t)n;
    const char *names[3] = {"x", "y", "z"};
    for (int f = 0; f < 3; ++f) {
        put(out->field_name[f], sizeof(out->field_name[f]), names[f]);
        out->field_offset[f] = 4u * f;
        out->field_datatype[f] = LMCMA_ROS_FLOAT32;
        out->field_count[f] = 1;
    }
    out->is_bigendian = 0;
    out->point_step = 16;  // pcl::PointXYZ: three floats padded to four (pcl::toROSMsg, S/:1065)
    out->row_step = out->point_step * out->width;
    out->is_dense = 1;
    out->data_bytes = (size_t)out->row_step;
    if (!data) return 0;
    if (data_capacity < out->data_bytes) return -4;
    const double o[3] = {origin ? origin[0] : 0.0, origin ? origin[1] : 0.0, origin ? origin[2] : 0.0};
    uint8_t *p = data;
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                if (!occ[((size_t)z * ny + y) * nx + x]) continue;
                const float v[4] = {(float)((x + 0.5) * cell_size + o[0]), (float)((y + 0.5) * cell_size + o[1]),
                                    (float)((z + 0.5) * cell_size + o[2]), 1.0f};
                memcpy(p, v, sizeof(v));
                p += sizeof(v);
            }
    out->data = data;
    return 0;
}

int lmcma_ros_grid_octomap_binary(const uint8_t *occ, int nx, int ny, int nz, double resolution,
                                  const int *key_origin, int8_t *data, size_t data_capacity,
                                  lmcma_ros_octomap *out)
{
    if (!occ || !out || nx < 1 || ny < 1 || nz < 1 || !(resolution > 0.0)) return -1;
    const int n[3] = {nx, ny, nz};
    int k0[3];
    for (int d = 0; d < 3; ++d) {
        k0[d] = kTreeMaxVal + (key_origin ? key_origin[d] : 0);
        if (k0[d] < 0 || (long)k0[d] + n[d] > 2L * kTreeMaxVal) return -1;  // does not fit the tree
    }
    std::vector<Level> lv(kTreeDepth + 1);
    for (int l = 0; l <= kTreeDepth; ++l) {
        for (int d = 0; d < 3; ++d) {
            lv[l].lo[d] = k0[d] >> l;
            lv[l].cnt[d] = ((k0[d] + n[d] - 1) >> l) - lv[l].lo[d] + 1;
        }
        lv[l].status.resize((size_t)lv[l].cnt[0] * lv[l].cnt[1] * lv[l].cnt[2]);
    }
    for (size_t c = 0; c < lv[0].status.size(); ++c) lv[0].status[c] = occ[c] ? kOccupied : kFree;
    for (int l = 1; l <= kTreeDepth; ++l) {
        Level &up = lv[l];
        const Level &dn = lv[l - 1];
        for (int z = 0; z < up.cnt[2]; ++z)
            for (int y = 0; y < up.cnt[1]; ++y)
                for (int x = 0; x < up.cnt[0]; ++x) {
                    const int ix = x + up.lo[0], iy = y + up.lo[1], iz = z + up.lo[2];
                    int seen[4] = {0, 0, 0, 0};
                    for (int i = 0; i < 8; ++i)
                        ++seen[dn.at(2 * ix + (i & 1), 2 * iy + ((i >> 1) & 1), 2 * iz + ((i >> 2) & 1))];
                    // eight equal leaves collapse into one (OcTree::prune)
                    const uint8_t s = seen[kFree] == 8 ? kFree : seen[kOccupied] == 8 ? kOccupied
                                      : seen[kUnknown] == 8 ? kUnknown : kInner;
                    up.status[((size_t)z * up.cnt[1] + y) * up.cnt[0] + x] = s;
                }
    }
    memset(out, 0, sizeof(*out));
    put(out->frame_id, sizeof(out->frame_id), LMCMA_ROS_FRAME);   // S/:1080-1082
    out->binary = 1;
    put(out->id, sizeof(out->id), "OcTree");
    out->resolution = resolution;
    size_t bytes = 0;
    // The root is written as an inner node whatever it holds (a grid that fills the whole key
    // space and is uniform would make it a leaf; such a grid does not fit the checks above
    // unless it has 65536 cells per side, which no caller has).
    if (lv[kTreeDepth].at(0, 0, 0) != kInner) return -1;
    emit(lv, kTreeDepth, 0, 0, 0, data, data_capacity, bytes);
    out->data_bytes = bytes;
    if (!data) return 0;
    if (data_capacity < bytes) return -4;
    out->data = data;
    return 0;
}

}  // extern "C"
