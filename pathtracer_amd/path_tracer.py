"""Host-side mirror of the reference's PathTracer class (src/path_tracer.hpp:11-19) on top of the C-ABI.

    tracer = PathTracer()
    tracer.Render(scene, samplesPerPixelIteration)     # src/path_tracer.cpp:158-222
    tracer.SaveImage(filename)                         # src/path_tracer.cpp:224-227

`scene` is a Scene: the host scene (pathtracer_amd.host.HostScene or a SceneData) together with its
samples-per-pixel list, the counterpart of the reference's Scene object.  Render uploads the scene once per
Scene, renders every sample of the chosen entry of the samples-per-pixel list and leaves the tonemapped image
in `renderedImage` (H x W x 3 float32, row-major like Image::m_pixels, src/image.cpp:105).

Several GPUs: one process per GPU; each process renders its share of the sample range into its own film and
the films are summed onto rank 0 with one torch.distributed reduce (RCCL over xGMI with the nccl backend).
"""
import numpy as np

from . import abi, api, host
from .scene import SceneData


class Scene:
    """What the reference's Scene hands to Render: flattened scene + samples-per-pixel list + output name."""

    def __init__(self, source, spp_list=None, output_filename=None):
        if isinstance(source, host.HostScene):
            info = source.info()
            self._desc = source.desc()
            self.numSamplesPerPixel = list(spp_list or info["spp"])
            self.outputImageFilename = output_filename or info["output_filename"]
        elif isinstance(source, SceneData):
            self._desc = source.to_desc()
            self.numSamplesPerPixel = list(spp_list or [32])          # src/scene.hpp:36
            self.outputImageFilename = output_filename or "rendered.png"  # src/scene.hpp:32
        else:
            raise TypeError("Scene needs a HostScene or a SceneData")
        self._source = source
        self._device_scene = None
        self._device = None

    @classmethod
    def Load(cls, filename, resource_dir):
        """Scene::Load (src/scene.cpp:294-368)."""
        return cls(host.HostScene.from_file(filename, resource_dir))

    def device_scene(self, device=0):
        if self._device_scene is None or self._device != device:
            if self._device_scene is not None:
                self._device_scene.close()
            self._desc.device = device
            self._device_scene = api.DeviceScene(self._desc)
            self._device = device
        return self._device_scene

    @property
    def imageResolution(self):
        return self._desc.image_width, self._desc.image_height


def sample_range(spp, rank, world_size):
    """Samples [begin, end) of every pixel that `rank` of `world_size` renders: contiguous, sizes differ by at
    most one, earlier ranks take the larger shares."""
    base, extra = divmod(spp, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class PathTracer:
    # A tree this small sits whole in the LDS copy the binary-tree kernels keep: its shadow rays are then fastest over the
    # binary tree (Cornell box, 256 samples per pixel: 67.9 ms against 69.4 over the eight-wide layout), its closest hits over
    # the eight-wide layout all the same (49.7 ms against 55.4 in the reference's order and 62.3 front to back)
    SMALL_TREE_NODES = 2400

    def __init__(self, device=0, seed=0, tonemap=abi.PT_TONEMAP_UNCHARTED2, ordered=None, traversal=None):
        self.device = device
        self.seed = seed
        self.tonemap = tonemap
        # "reference", "ordered" (front to back over the binary tree) or "wide" (the eight-wide layout, shadow rays too): the
        # same picture from each; None: "wide" unless the tree is small.  `ordered` is the older switch: True = "ordered",
        # False = "reference".
        self.traversal = traversal if traversal is not None else {None: None, True: "ordered", False: "reference"}[ordered]
        self.renderedImage = None   # tonemapped, what the reference keeps (src/path_tracer.cpp:210-221)
        self.linearImage = None     # the image before the tonemap pass, which the reference overwrites
        self.stats = None

    def Render(self, scene, samplesPerPixelIteration=0, process_group=None):
        """Renders numSamplesPerPixel[samplesPerPixelIteration] samples per pixel.

        With torch.distributed initialised (or a process_group given) the sample range is split over the
        ranks, the films are reduced onto rank 0 and only rank 0 ends up with the images.
        """
        spp = scene.numSamplesPerPixel[samplesPerPixelIteration]
        dev = scene.device_scene(self.device)
        rank, world = 0, 1
        dist = None
        try:
            import torch.distributed as dist_mod
            if dist_mod.is_available() and dist_mod.is_initialized():
                dist = dist_mod
                rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        except ImportError:
            pass
        begin, end = sample_range(spp, rank, world)
        traversal = self.traversal or "wide"
        wide_shadows = traversal == "wide" and (self.traversal == "wide" or scene._desc.num_nodes > self.SMALL_TREE_NODES)
        try:
            dev.render(spp, begin, end, seed=self.seed, ordered=traversal != "reference", wide=traversal == "wide", wide_shadows=wide_shadows)
        except api.PtError as e:
            if self.traversal is not None or e.code != abi.PT_ERR_STATE:
                raise
            # a tree the eight-wide layout cannot express (a box that is not finite or lies outside its parent's)
            small = scene._desc.num_nodes <= self.SMALL_TREE_NODES
            dev.render(spp, begin, end, seed=self.seed, ordered=not small)
        if world > 1:
            reduce_film(dev, self.device, dist, process_group)
            if rank != 0:
                self.renderedImage = self.linearImage = None
                self.stats = dev.stats()
                return
        self.linearImage, self.renderedImage, self.stats = dev.readback(self.tonemap)

    def SaveImage(self, filename):
        """Image::Save (src/image.cpp:43-100): 8-bit quantisation of the tonemapped image."""
        if self.renderedImage is None:
            return False
        host.write_image(filename, self.renderedImage)
        return True


class _DevicePointer:
    """Exposes a raw device allocation through __cuda_array_interface__ so that torch can wrap it."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3}


def film_tensor(dev, device_index):
    """The scene's float4 film as a torch tensor sharing the library's device memory."""
    import torch
    ptr, nbytes = dev.film_device_ptr()
    return torch.as_tensor(_DevicePointer(ptr, nbytes // 4), device=f"cuda:{device_index}")


def reduce_film(dev, device_index, dist, process_group=None):
    """Sum of the per-GPU films (sum r, g, b and sample count per pixel) onto rank 0: one collective.  With the nccl
    backend that is one RCCL reduce on the device buffers over xGMI.  The gloo backend (rehearsals of the
    multi-process path on a machine with one GPU, tests) reduces host tensors, so the film goes through host memory."""
    import torch
    film = film_tensor(dev, device_index)
    if dist.get_backend(process_group) == "gloo":
        staged = film.cpu()
        dist.reduce(staged, dst=0, op=dist.ReduceOp.SUM, group=process_group)
        if dist.get_rank(process_group) == 0:
            film.copy_(staged)
    else:
        dist.reduce(film, dst=0, op=dist.ReduceOp.SUM, group=process_group)
    torch.cuda.synchronize(device_index)
