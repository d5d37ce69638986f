"""MI355X-native voxel ray traversal: the LSVO::castRay / RayCaster hot path of
johnBuffer/CpuVoxelRaycaster as hand-written HIP kernels behind a C ABI
(include/vrc.h).  See DESIGN.md."""
from . import capi
from .capi import VrcError, HIT_DTYPE, LNODE_DTYPE, build_terrain_lsvo, build_volume_lsvo, make_rotation
from .raycaster import LSVO, Grid3D, RayCaster, Presenter, make_camera
from .scenes import terrain_heights, load_textures, load_bmp, load_textures_bmp, reference_camera, reference_camera_position, reference_light

__all__ = ["capi", "VrcError", "HIT_DTYPE", "LNODE_DTYPE", "build_terrain_lsvo", "build_volume_lsvo",
           "make_rotation", "LSVO", "Grid3D", "RayCaster", "Presenter", "make_camera", "terrain_heights", "load_bmp", "load_textures_bmp",
           "load_textures", "reference_camera", "reference_camera_position", "reference_light"]
