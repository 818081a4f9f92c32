"""The step after the path: the converged path and the map laid out as the ROS messages
the reference's node publishes (include/lmcma_ros.h).  Host code only, no GPU, no ROS.

Topic names, frame, namespace and id are those of
sample_based_optimisation_based_path_planner.cpp:1063, 1079, 1470-1482.  The octree
stream is checked against hand-computed bytes and by decoding it back to the grid with
an independent reader of the published OcTree binary format (octomap itself is not
part of the reference's sources: parity unpinned).
"""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_topic_names_and_frame_are_the_reference_ones():
    from lmcma_path_planner_amd import rosmsg
    text = open(os.path.join(ROOT, "include", "lmcma_ros.h")).read()
    defs = dict(re.findall(r'#define (LMCMA_ROS_\w+) "([^"]+)"', text))
    assert defs["LMCMA_ROS_TOPIC_MARKER"] == rosmsg.TOPIC_MARKER == "visualization_marker"
    assert defs["LMCMA_ROS_TOPIC_POINTCLOUD2"] == rosmsg.TOPIC_POINTCLOUD2 == "pointcloud2"
    assert defs["LMCMA_ROS_TOPIC_OCTOMAP"] == rosmsg.TOPIC_OCTOMAP == "octomapbinary"
    assert defs["LMCMA_ROS_FRAME"] == rosmsg.FRAME == "base_link"


def test_struct_layouts_match_the_header():
    from lmcma_path_planner_amd import _lib
    # marker: 2 x char[32], 3 int32 (+4 pad), 10 doubles, 4 floats, int32 (+4 pad), pointer
    assert ctypes.sizeof(_lib.RosMarker) == 64 + 16 + 80 + 16 + 8 + 8
    assert _lib.RosMarker.points.offset == 184
    assert _lib.RosPointCloud2.data.offset == ctypes.sizeof(_lib.RosPointCloud2) - 8
    assert _lib.RosOctomap.resolution.offset == 56


def test_path_marker_2d():
    from lmcma_path_planner_amd import rosmsg
    W = 5
    x = np.concatenate([np.arange(W) + 10.0, np.arange(W) + 20.0])   # [x0..x4 | y0..y4]
    m = rosmsg.path_marker(x, 2, W, start=(99.0, 0.0), goal=(0.0, 99.0), cell_size=0.1, line_width=0.02)
    assert m["header"]["frame_id"] == "base_link" and m["ns"] == "my_namespace" and m["id"] == 1
    assert m["type"] == rosmsg.LINE_STRIP and m["action"] == rosmsg.ADD
    assert m["pose"]["orientation"] == [0.0, 0.0, 0.0, 1.0]
    assert m["scale"][0] == 0.02 and m["color"][3] == 1.0
    pts = m["points"]
    assert pts.shape == (W + 2, 3)
    np.testing.assert_allclose(pts[0], [9.9, 0.0, 0.0])
    np.testing.assert_allclose(pts[-1], [0.0, 9.9, 0.0])
    np.testing.assert_allclose(pts[1:-1, 0], (np.arange(W) + 10.0) * 0.1)
    np.testing.assert_allclose(pts[1:-1, 1], (np.arange(W) + 20.0) * 0.1)
    assert np.all(pts[:, 2] == 0.0)


def test_path_marker_3d_and_delete_and_errors():
    from lmcma_path_planner_amd import _lib, rosmsg
    W = 3
    x = np.arange(3 * W, dtype=np.float64)
    m = rosmsg.path_marker(x, 3, W, start=(1, 2, 3), goal=(4, 5, 6))
    np.testing.assert_array_equal(m["points"], [[1, 2, 3], [0, 3, 6], [1, 4, 7], [2, 5, 8], [4, 5, 6]])
    d = rosmsg.path_marker_delete()
    assert d["action"] == rosmsg.DELETE and d["points"].shape == (0, 3) and d["id"] == 1
    # capacity too small, bad dims
    lib = _lib.load()
    pts = np.zeros((W + 1, 3))
    s = np.zeros(3)
    mk = _lib.RosMarker()
    dp = _lib.c_double_p
    assert lib.lmcma_ros_path_marker(x.ctypes.data_as(dp), 3, W, s.ctypes.data_as(dp), s.ctypes.data_as(dp), 1.0,
                                     0.1, pts.ctypes.data_as(dp), W + 1, ctypes.byref(mk)) == -4
    assert lib.lmcma_ros_path_marker(x.ctypes.data_as(dp), 4, W, s.ctypes.data_as(dp), s.ctypes.data_as(dp), 1.0,
                                     0.1, pts.ctypes.data_as(dp), W + 2, ctypes.byref(mk)) == -1


def test_pointcloud2_layout_and_points():
    from lmcma_path_planner_amd import grids, rosmsg
    occ = grids.two_bar_map(100)
    pc = rosmsg.grid_pointcloud2(occ, cell_size=0.01)
    n = int(np.count_nonzero(occ))
    assert pc["header"]["frame_id"] == "base_link"
    assert pc["height"] == 1 and pc["width"] == n
    assert [f["name"] for f in pc["fields"]] == ["x", "y", "z"]
    assert [f["offset"] for f in pc["fields"]] == [0, 4, 8]
    assert all(f["datatype"] == rosmsg.FLOAT32 and f["count"] == 1 for f in pc["fields"])
    assert pc["point_step"] == 16 and pc["row_step"] == 16 * n and not pc["is_bigendian"] and pc["is_dense"]
    assert pc["data"].size == 16 * n
    xyz = pc["data"].view(np.float32).reshape(n, 4)
    ys, xs = np.nonzero(occ)
    np.testing.assert_allclose(xyz[:, 0], (xs + 0.5) * 0.01, rtol=1e-6)
    np.testing.assert_allclose(xyz[:, 1], (ys + 0.5) * 0.01, rtol=1e-6)
    np.testing.assert_allclose(xyz[:, 2], 0.005, rtol=1e-6)
    # an empty map is an empty cloud
    pc0 = rosmsg.grid_pointcloud2(np.zeros((4, 4), np.uint8))
    assert pc0["width"] == 0 and pc0["data"].size == 0


# ---- an independent reader of the OcTree binary stream ---------------------------------

def decode_octree(data, depth=16):
    """Leaves of the stream as {(key_x, key_y, key_z, size): occupied}; keys are the low corner."""
    data = np.asarray(data).view(np.uint8)
    leaves = {}
    pos = [0]

    def node(level, kx, ky, kz):
        bits = int(data[pos[0]]) | (int(data[pos[0] + 1]) << 8)
        pos[0] += 2
        half = 1 << (level - 1)
        inner = []
        for i in range(8):
            two = (bits >> (2 * i)) & 3
            cx, cy, cz = kx + half * (i & 1), ky + half * ((i >> 1) & 1), kz + half * ((i >> 2) & 1)
            if two == 1:
                leaves[(cx, cy, cz, half)] = False
            elif two == 2:
                leaves[(cx, cy, cz, half)] = True
            elif two == 3:
                assert level > 1, "an inner node below the last level"
                inner.append((cx, cy, cz))
        for (cx, cy, cz) in inner:
            node(level - 1, cx, cy, cz)

    node(depth, 0, 0, 0)
    assert pos[0] == data.size, "trailing bytes in the stream"
    return leaves


def grid_from_leaves(leaves, shape, key0):
    nz, ny, nx = shape
    g = np.full(shape, 255, dtype=np.uint8)   # 255 = unknown
    for (kx, ky, kz, size), occ in leaves.items():
        x0, y0, z0 = kx - key0[0], ky - key0[1], kz - key0[2]
        assert 0 <= x0 and x0 + size <= nx and 0 <= y0 and y0 + size <= ny and 0 <= z0 and z0 + size <= nz, \
            "a leaf outside the grid"
        g[z0:z0 + size, y0:y0 + size, x0:x0 + size] = 1 if occ else 0
    return g


def test_octomap_single_cell_known_answer():
    from lmcma_path_planner_amd import rosmsg
    # one occupied cell at key (32768, 32768, 32768): the first cell of the upper octant.
    # Root: child 7 is inner -> bits 14,15 -> bytes 00 C0; 14 more inner nodes whose child 0 is
    # inner -> 03 00; the last node's child 0 is the occupied leaf -> 02 00.
    om = rosmsg.grid_octomap_binary(np.ones((1, 1, 1), np.uint8), 0.05)
    assert om["header"]["frame_id"] == "base_link" and om["binary"] and om["id"] == "OcTree"
    assert om["resolution"] == 0.05
    want = [0x00, 0xC0] + [0x03, 0x00] * 14 + [0x02, 0x00]
    assert list(om["data"].view(np.uint8)) == want
    # a free cell one to the left in x: key (32767, 32768, 32768) is in root child 6, last cell in x
    om = rosmsg.grid_octomap_binary(np.zeros((1, 1, 1), np.uint8), 0.05, key_origin=(-1, 0, 0))
    want = [0x00, 0x30] + [0x0C, 0x00] * 14 + [0x04, 0x00]
    assert list(om["data"].view(np.uint8)) == want


@pytest.mark.parametrize("shape,origin", [((1, 100, 100), (0, 0, 0)), ((12, 20, 9), (-5, 3, -7)),
                                          ((16, 16, 16), (0, 0, 0)), ((8, 8, 8), (-8, -8, -8))])
def test_octomap_stream_decodes_back_to_the_grid(shape, origin):
    from lmcma_path_planner_amd import grids, rosmsg
    rng = np.random.default_rng(5)
    if shape[0] == 1:
        occ = grids.two_bar_map(100)[None]
    else:
        occ = (rng.random(shape) < 0.3).astype(np.uint8)
        occ[: shape[0] // 2, : shape[1] // 2, : shape[2] // 2] = 1    # a block that prunes
    om = rosmsg.grid_octomap_binary(occ, 0.1, key_origin=origin)
    leaves = decode_octree(om["data"])
    key0 = tuple(32768 + o for o in origin)
    back = grid_from_leaves(leaves, occ.shape, key0)
    np.testing.assert_array_equal(back, (occ != 0).astype(np.uint8))   # every cell known, same state
    # pruned: no eight sibling leaves of one state remain
    by_parent = {}
    for (kx, ky, kz, size), o in leaves.items():
        p = (kx // (2 * size), ky // (2 * size), kz // (2 * size), size)
        by_parent.setdefault(p, []).append(o)
    assert not any(len(v) == 8 and len(set(v)) == 1 for v in by_parent.values())


def test_octomap_uniform_aligned_block_is_one_leaf():
    from lmcma_path_planner_amd import rosmsg
    om = rosmsg.grid_octomap_binary(np.ones((8, 8, 8), np.uint8), 1.0)
    leaves = decode_octree(om["data"])
    assert leaves == {(32768, 32768, 32768, 8): True}
    # 16 levels: root + 12 inner nodes down to the parent of the size-8 leaf
    assert om["data"].size == 2 * 13


def test_octomap_rejects_grids_outside_the_tree():
    from lmcma_path_planner_amd import _lib, rosmsg
    with pytest.raises(_lib.LmcmaError):
        rosmsg.grid_octomap_binary(np.ones((1, 1, 4), np.uint8), 1.0, key_origin=(32766, 0, 0))


def test_ros_node_source_uses_the_header_names():
    """The catkin-guarded node publishes on the header's topics; it is compiled only where ROS exists."""
    src = open(os.path.join(ROOT, "ros", "lmcma_planner_node.cpp")).read()
    for name in ("LMCMA_ROS_TOPIC_MARKER", "LMCMA_ROS_TOPIC_POINTCLOUD2", "LMCMA_ROS_TOPIC_OCTOMAP",
                 "lmcma_ros_path_marker", "lmcma_ros_grid_pointcloud2", "lmcma_ros_grid_octomap_binary"):
        assert name in src
    cm = open(os.path.join(ROOT, "ros", "CMakeLists.txt")).read()
    assert "find_package(catkin" in cm and "QUIET" in cm
