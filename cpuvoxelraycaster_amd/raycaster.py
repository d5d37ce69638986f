"""Thin Python mirror of the reference's operator interface for the hot path,
over the C ABI (include/vrc.h).  Names follow the reference:

  LSVO.castRay                 <- LSVO<N>::castRay          (include/lsvo.hpp:33)
  RayCaster.setLightPosition   <- RayCaster::setLightPosition (include/raycaster.hpp:62)
  RayCaster.renderFrame        <- the swarm lambda: getRay + renderRay per pixel (src/main.cpp:139-152)
  RayCaster.samples_to_image   <- raycaster.hpp:94
  RayCaster.resetSamples       <- raycaster.hpp:105
  Presenter.present            <- the SFML blend / upscale chain after the frame (src/main.cpp:160-182)

All compute runs in the HIP kernels of libvrc_hip.so; numpy / torch are only
used to hold buffers."""
import ctypes as C

import numpy as np

from . import capi
from .capi import Camera, FrameParams, FrameStats, HIT_DTYPE, VrcError, check, ptr


class LSVO:
    """Device-resident linear sparse voxel octree (Volumetric implementation)."""

    def __init__(self, lnodes, depth, device=0, textures=None):
        L = capi.load()
        lnodes = np.ascontiguousarray(lnodes)
        if lnodes.dtype.itemsize != 8:
            raise VrcError("lnodes must be an array of 8-byte LNode records")
        self._h = C.c_void_p()
        check(L.vrc_scene_create(ptr(lnodes), lnodes.shape[0], depth, device, C.byref(self._h)))
        self.depth = depth
        self.device = device
        self.n_nodes = int(lnodes.shape[0])
        if textures is not None:
            self.setTextures(*textures)

    @classmethod
    def _from_handle(cls, handle, depth, device, textures):
        self = cls.__new__(cls)
        self._h = handle
        self.depth, self.device = depth, device
        self.n_nodes = int(capi.load().vrc_scene_node_count(handle))
        self.build_ms = None
        if textures is not None:
            self.setTextures(*textures)
        return self

    @classmethod
    def fromTerrain(cls, height_i32, depth, device=0, textures=None):
        """Build the LSVO of the reference's terrain generator on the GPU (main.cpp:59-88)."""
        size = 1 << depth
        h = np.ascontiguousarray(np.asarray(height_i32)[:size, :size], dtype=np.int32)
        handle, ms = C.c_void_p(), C.c_float()
        check(capi.load().vrc_scene_build_terrain(ptr(h), depth, device, C.byref(handle), C.byref(ms)))
        self = cls._from_handle(handle, depth, device, textures)
        self.build_ms = ms.value
        return self

    @classmethod
    def fromFastNoiseTerrain(cls, depth, seed=1337, device=0, textures=None):
        """main.cpp:59-88 end to end on the GPU: FastNoise heights -> LSVO."""
        handle, ms = C.c_void_p(), C.c_float()
        check(capi.load().vrc_scene_build_fastnoise_terrain(seed, depth, device, C.byref(handle), C.byref(ms)))
        self = cls._from_handle(handle, depth, device, textures)
        self.build_ms = ms.value
        return self

    @classmethod
    def fromVolume(cls, solid_u8, depth, device=0, textures=None):
        """Build the LSVO of an arbitrary occupancy volume solid[x, y, z] on the GPU."""
        size = 1 << depth
        s = np.ascontiguousarray(solid_u8, dtype=np.uint8)
        assert s.shape == (size, size, size)
        handle, ms = C.c_void_p(), C.c_float()
        check(capi.load().vrc_scene_build_volume(ptr(s), depth, device, C.byref(handle), C.byref(ms)))
        self = cls._from_handle(handle, depth, device, textures)
        self.build_ms = ms.value
        return self

    def downloadNodes(self):
        out = np.zeros(self.n_nodes, capi.LNODE_DTYPE)
        check(capi.load().vrc_scene_download_nodes(self._h, ptr(out)))
        return out

    def setTextures(self, top_rgb, side_rgb):
        top = np.ascontiguousarray(top_rgb, dtype=np.uint8).reshape(-1)
        side = np.ascontiguousarray(side_rgb, dtype=np.uint8).reshape(-1)
        assert top.size == 768 and side.size == 768
        check(capi.load().vrc_scene_set_textures(self._h, ptr(top), ptr(side)))

    def castRay(self, position, direction, ray_size_coef=0.0, ray_size_bias=0.0):
        """Single ray (Camera::getClosestPoint path); returns a HIT_DTYPE record."""
        out = np.zeros(1, HIT_DTYPE)
        o = np.ascontiguousarray(position, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        check(capi.load().vrc_cast_ray(self._h, ptr(o), ptr(d), ray_size_coef, ray_size_bias, ptr(out)))
        return out[0]

    def castRays(self, org, dir_, coef=None, bias=None):
        """Batch form over host arrays."""
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dir_ = np.ascontiguousarray(dir_, np.float32).reshape(-1, 3)
        n = org.shape[0]
        if coef is not None:
            coef = np.ascontiguousarray(np.broadcast_to(np.asarray(coef, np.float32), (n,)))
        if bias is not None:
            bias = np.ascontiguousarray(np.broadcast_to(np.asarray(bias, np.float32), (n,)))
        out = np.zeros(n, HIT_DTYPE)
        check(capi.load().vrc_cast_rays(self._h, n, ptr(org), ptr(dir_), ptr(coef), ptr(bias), ptr(out),
                                        capi.VRC_MEM_HOST, None))
        return out

    def castRaysDevice(self, n, org_ptr, dir_ptr, out_ptr, coef_ptr=None, bias_ptr=None, stream=None):
        """Batch form over device pointers (asynchronous on `stream`)."""
        check(capi.load().vrc_cast_rays(self._h, n, ptr(org_ptr), ptr(dir_ptr), ptr(coef_ptr), ptr(bias_ptr),
                                        ptr(out_ptr), capi.VRC_MEM_DEVICE, ptr(stream)))

    def castRayChainsDevice(self, n, org_a_ptr, dir_a_ptr, org_b_ptr, dir_b_ptr, coef_b, out_a_ptr, out_b_ptr, not_executed_ptr=None, stream=None):
        """ray A from the root, then ray B started below the root next to A's hit (include/vrc.h: vrc_cast_ray_chains)"""
        check(capi.load().vrc_cast_ray_chains(self._h, n, ptr(org_a_ptr), ptr(dir_a_ptr), ptr(org_b_ptr), ptr(dir_b_ptr), float(coef_b),
                                              ptr(out_a_ptr), ptr(out_b_ptr), ptr(not_executed_ptr), ptr(stream)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.load().vrc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Grid3D:
    """Dense grid (grid_3d.hpp); cells[x, y, z] = Cell::Type (0 = Empty)."""

    def __init__(self, cells, device=0):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        X, Y, Z = cells.shape
        self._h = C.c_void_p()
        check(capi.load().vrc_grid_create(ptr(cells), X, Y, Z, device, C.byref(self._h)))

    def castRays(self, org, dir_):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dir_ = np.ascontiguousarray(dir_, np.float32).reshape(-1, 3)
        out = np.zeros(org.shape[0], HIT_DTYPE)
        check(capi.load().vrc_grid_cast_rays(self._h, org.shape[0], ptr(org), ptr(dir_), ptr(out),
                                             capi.VRC_MEM_HOST, None))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.load().vrc_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_camera(position, rot, fov=1.0, aperture=0.0, focal_length=1.0):
    cam = Camera()
    cam.position[:] = [float(v) for v in position]
    cam.rot[:] = [float(v) for v in rot]
    cam.fov, cam.aperture, cam.focal_length = fov, aperture, focal_length
    return cam


class RayCaster:
    """raycaster.hpp:43-283 with the framebuffer and sample accumulators on the GPU."""

    def __init__(self, svo, render_size):
        self.svo = svo
        self.width, self.height = int(render_size[0]), int(render_size[1])
        self._h = C.c_void_p()
        check(capi.load().vrc_renderer_create(svo._h, self.width, self.height, C.byref(self._h)))
        self.light_position = (0.0, 0.0, 0.0)
        self.use_gi = False
        self.use_samples = False
        self.shadow_samples = 0      # 0 = reference default
        self.gi_bounces = 1
        self.seed = 0x9E3779B9
        self.frame_index = 0

    def setLightPosition(self, position):
        self.light_position = tuple(float(v) for v in position)

    def params(self, spp=1, checker_parity=-1, row_block=0, shard_index=0, shard_count=1):
        p = FrameParams()
        p.light_position[:] = self.light_position
        p.use_gi, p.use_samples = int(self.use_gi), int(self.use_samples)
        p.shadow_samples, p.gi_bounces = self.shadow_samples, self.gi_bounces
        p.checker_parity, p.spp = checker_parity, spp
        p.seed, p.frame_index = self.seed, self.frame_index
        p.row_block, p.shard_index, p.shard_count = row_block, shard_index, shard_count
        return p

    def renderFrame(self, camera, spp=1, checker_parity=-1, stream=None, row_block=0, shard_index=0, shard_count=1):
        p = self.params(spp, checker_parity, row_block, shard_index, shard_count)
        check(capi.load().vrc_render_frame(self._h, C.byref(camera), C.byref(p), ptr(stream)))
        self.frame_index += spp

    # scheduling knobs of THIS renderer (results never depend on them; see include/vrc.h)
    def setTuning(self, blocks_per_cu=0):
        check(capi.load().vrc_renderer_set_tuning(self._h, blocks_per_cu))

    def setSampleChunk(self, samples_per_unit):
        check(capi.load().vrc_renderer_set_sample_chunk(self._h, samples_per_unit))

    def lastKernel(self):
        """symbol of the frame kernel the last renderFrame* launched (what a rocprofv3 trace of the run lists)"""
        return capi.load().vrc_renderer_last_kernel(self._h).decode()

    def setWalkFromRoot(self, on=True):
        """Measurement switch: every ray from the root (none starts below it).  include/vrc.h: vrc_renderer_set_walk_from_root."""
        check(capi.load().vrc_renderer_set_walk_from_root(self._h, 1 if on else 0))

    def setLaneSamples(self, samples=0):
        """lane <-> (pixel, sample) map of the frame kernel: 1 = 8 x 8 pixels per wave, 4 = 4 x 4 pixels x 4 samples abreast,
        0 = the library's choice.  Same results either way.  include/vrc.h: vrc_renderer_set_lane_samples."""
        check(capi.load().vrc_renderer_set_lane_samples(self._h, samples))

    def setQuadWalks(self, on=True):
        """pinhole camera: the sample-invariant primary / shadow walks one quadrant of the tile at a time, four samples abreast (on
        by default where a launch allows it).  Same results.  include/vrc.h: vrc_renderer_set_quad_walks."""
        check(capi.load().vrc_renderer_set_quad_walks(self._h, 1 if on else 0))

    def setInvariantRayReuse(self, on=True):
        """beyond the reference: pinhole camera, walk a work unit's primary and shadow ray once instead of once per sample
        (same image; stats then count the walks executed).  include/vrc.h: vrc_renderer_set_invariant_ray_reuse."""
        check(capi.load().vrc_renderer_set_invariant_ray_reuse(self._h, 1 if on else 0))

    def renderFrameResolved(self, camera, spp=1, dst_ptr=None, stream=None, row_block=0, shard_index=0, shard_count=1):
        """renderFrame + resolveShard(..., reset=True) in one launch (vrc_render_frame_resolved): the frame's samples
        rendered, resolved into the image (and the packed shard buffer dst_ptr), accumulators left at zero."""
        p = self.params(spp, -1, row_block, shard_index, shard_count)
        check(capi.load().vrc_render_frame_resolved(self._h, C.byref(camera), C.byref(p), ptr(dst_ptr), ptr(stream)))
        self.frame_index += spp

    # ---- direct peer writes (include/vrc.h: vrc_ipc_*, vrc_renderer_set_image_target) ----
    def exportImage(self):
        """64 opaque bytes naming this renderer's framebuffer for the other processes of the node"""
        h = (C.c_ubyte * 64)()
        check(capi.load().vrc_ipc_export_image(self._h, C.byref(h)))
        return bytes(h)

    def setImageTarget(self, image_dev_ptr):
        """sharded sample-mode frames are resolved into that framebuffer (full-frame layout) instead of this renderer's own;
        None = own image again"""
        check(capi.load().vrc_renderer_set_image_target(self._h, ptr(image_dev_ptr)))

    def setPrimaryCapture(self, dev_ptr):
        check(capi.load().vrc_renderer_set_primary_capture(self._h, ptr(dev_ptr)))

    def samples_to_image(self, stream=None):
        check(capi.load().vrc_samples_to_image(self._h, ptr(stream)))

    def resetSamples(self, stream=None):
        check(capi.load().vrc_reset_samples(self._h, ptr(stream)))

    def clearImage(self, stream=None):
        check(capi.load().vrc_clear_image(self._h, ptr(stream)))

    def image_ptr(self):
        return capi.load().vrc_image_device_ptr(self._h)

    def readImage(self, stream=None):
        img = np.zeros((self.height, self.width, 4), np.uint8)
        check(capi.load().vrc_read_image(self._h, ptr(img), ptr(stream)))
        return img

    def writeImage(self, img, stream=None):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.height, self.width, 4)
        check(capi.load().vrc_write_image(self._h, ptr(img), ptr(stream)))

    def readAccum(self, stream=None):
        acc = np.zeros((self.height, self.width, 4), np.uint32)
        check(capi.load().vrc_read_accum(self._h, ptr(acc), ptr(stream)))
        return acc

    def stats(self, reset=False, stream=None):
        st = FrameStats()
        check(capi.load().vrc_get_stats(self._h, C.byref(st), int(reset), ptr(stream)))
        return st

    def packShard(self, row_block, shard_index, shard_count, dst_ptr, stream=None):
        check(capi.load().vrc_pack_shard(self._h, row_block, shard_index, shard_count, ptr(dst_ptr), ptr(stream)))

    def resolveShard(self, row_block=0, shard_index=0, shard_count=1, dst_ptr=None, reset=False, stream=None):
        """samples_to_image (+ packShard, + resetSamples) for this shard's rows in one kernel."""
        check(capi.load().vrc_resolve_shard(self._h, row_block, shard_index, shard_count, ptr(dst_ptr), int(reset), ptr(stream)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.load().vrc_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Presenter:
    """main.cpp:160-182 on the device: render_tex / denoised_tex persistence blend, nearest upscale to the window size,
    optional median (res/median_3.frag, res/median.frag).  Owns denoised_tex and the window image."""

    def __init__(self, render_size, window_size, device=0):
        self.width, self.height = int(render_size[0]), int(render_size[1])
        self.out_width, self.out_height = int(window_size[0]), int(window_size[1])
        self._h = C.c_void_p()
        check(capi.load().vrc_presenter_create(device, self.width, self.height, self.out_width, self.out_height, C.byref(self._h)))

    def present(self, raycaster, old_value_conservation=None, median=0, stream=None):
        """old_value_conservation defaults to main.cpp:161: use_samples ? 0 : 0.1"""
        if old_value_conservation is None:
            old_value_conservation = 0.0 if raycaster.use_samples else 0.1
        check(capi.load().vrc_present(self._h, raycaster._h, old_value_conservation, median, ptr(stream)))

    def presentImage(self, image_dev_ptr, old_value_conservation, median=0, stream=None):
        check(capi.load().vrc_present_image(self._h, ptr(image_dev_ptr), old_value_conservation, median, ptr(stream)))

    def clear(self, stream=None):
        check(capi.load().vrc_presenter_clear(self._h, ptr(stream)))

    def window_ptr(self):
        return capi.load().vrc_presenter_window_ptr(self._h)

    def read(self, stream=None):
        """(window RGBA8 (out_h, out_w, 4), denoised_tex RGBA8 (h, w, 4))"""
        win = np.zeros((self.out_height, self.out_width, 4), np.uint8)
        den = np.zeros((self.height, self.width, 4), np.uint8)
        check(capi.load().vrc_presenter_read(self._h, ptr(win), ptr(den), ptr(stream)))
        return win, den

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            capi.load().vrc_presenter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
