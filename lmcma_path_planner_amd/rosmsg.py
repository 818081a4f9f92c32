"""The converged path and the map as the ROS messages the reference's node publishes
(include/lmcma_ros.h): plain dictionaries whose keys are the message fields, filled by
the library's host functions.  Nothing here needs ROS; a node copies the fields into
visualization_msgs/Marker, sensor_msgs/PointCloud2 and octomap_msgs/Octomap.

Topic names and frame are those of
sample_based_optimisation_based_path_planner.cpp:1063, 1079, 1470-1472.
"""
import ctypes

import numpy as np

from . import _lib

TOPIC_MARKER = "visualization_marker"
TOPIC_POINTCLOUD2 = "pointcloud2"
TOPIC_OCTOMAP = "octomapbinary"
FRAME = "base_link"
LINE_STRIP, ADD, DELETE, FLOAT32 = 4, 0, 2, 7


def _check(rc, what):
    if rc != 0:
        raise _lib.LmcmaError("%s failed (%d)" % (what, rc))


def _marker_dict(m, points):
    return {"header": {"frame_id": m.frame_id.decode()}, "ns": m.ns.decode(), "id": m.id, "type": m.type,
            "action": m.action, "pose": {"position": list(m.position), "orientation": list(m.orientation)},
            "scale": list(m.scale), "color": list(m.color), "points": points}


def path_marker(x, dims, waypoints, start, goal, cell_size=1.0, line_width=0.05):
    """LINE_STRIP through start, the waypoints of the dimension-major vector x, and goal."""
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.size < dims * waypoints:
        raise ValueError("x holds fewer than dims * waypoints values")
    s = np.zeros(3)
    g = np.zeros(3)
    s[:dims] = np.asarray(start, dtype=np.float64)[:dims]
    g[:dims] = np.asarray(goal, dtype=np.float64)[:dims]
    pts = np.zeros((waypoints + 2, 3))
    m = _lib.RosMarker()
    _check(lib.lmcma_ros_path_marker(x.ctypes.data_as(_lib.c_double_p), dims, waypoints,
                                     s.ctypes.data_as(_lib.c_double_p), g.ctypes.data_as(_lib.c_double_p),
                                     float(cell_size), float(line_width), pts.ctypes.data_as(_lib.c_double_p),
                                     waypoints + 2, ctypes.byref(m)), "lmcma_ros_path_marker")
    return _marker_dict(m, pts[:m.n_points].copy())


def path_marker_delete():
    lib = _lib.load()
    m = _lib.RosMarker()
    _check(lib.lmcma_ros_path_marker_delete(ctypes.byref(m)), "lmcma_ros_path_marker_delete")
    return _marker_dict(m, np.zeros((0, 3)))


def _occ3(occ):
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    while occ.ndim < 3:
        occ = occ[None]
    return occ


def grid_pointcloud2(occ, cell_size=1.0, origin=None):
    """Occupied cells of occ ([nz][ny][nx], [ny][nx] or [nx]) as a PointCloud2 of cell centres."""
    lib = _lib.load()
    occ = _occ3(occ)
    nz, ny, nx = occ.shape
    org = None if origin is None else np.ascontiguousarray(origin, dtype=np.float64)
    org_p = None if org is None else org.ctypes.data_as(_lib.c_double_p)
    pc = _lib.RosPointCloud2()
    _check(lib.lmcma_ros_grid_pointcloud2(occ.ctypes.data, nx, ny, nz, float(cell_size), org_p, None, 0,
                                          ctypes.byref(pc)), "lmcma_ros_grid_pointcloud2")
    data = np.zeros(max(pc.data_bytes, 1), dtype=np.uint8)
    _check(lib.lmcma_ros_grid_pointcloud2(occ.ctypes.data, nx, ny, nz, float(cell_size), org_p, data.ctypes.data,
                                          data.size, ctypes.byref(pc)), "lmcma_ros_grid_pointcloud2")
    fields = [{"name": pc.field_name[f].value.decode(), "offset": pc.field_offset[f],
               "datatype": pc.field_datatype[f], "count": pc.field_count[f]} for f in range(3)]
    return {"header": {"frame_id": pc.frame_id.decode()}, "height": pc.height, "width": pc.width, "fields": fields,
            "is_bigendian": bool(pc.is_bigendian), "point_step": pc.point_step, "row_step": pc.row_step,
            "is_dense": bool(pc.is_dense), "data": data[:pc.data_bytes].copy()}


def grid_octomap_binary(occ, resolution, key_origin=None):
    """occ as the binary OcTree stream of an octomap_msgs/Octomap."""
    lib = _lib.load()
    occ = _occ3(occ)
    nz, ny, nx = occ.shape
    ko = None if key_origin is None else (ctypes.c_int * 3)(*[int(v) for v in key_origin])
    om = _lib.RosOctomap()
    _check(lib.lmcma_ros_grid_octomap_binary(occ.ctypes.data, nx, ny, nz, float(resolution), ko, None, 0,
                                             ctypes.byref(om)), "lmcma_ros_grid_octomap_binary")
    data = np.zeros(max(om.data_bytes, 1), dtype=np.int8)
    _check(lib.lmcma_ros_grid_octomap_binary(occ.ctypes.data, nx, ny, nz, float(resolution), ko, data.ctypes.data,
                                             data.size, ctypes.byref(om)), "lmcma_ros_grid_octomap_binary")
    return {"header": {"frame_id": om.frame_id.decode()}, "binary": bool(om.binary), "id": om.id.decode(),
            "resolution": om.resolution, "data": data[:om.data_bytes].copy()}
