"""virconv_amd: MI355X-native (gfx950) Virtual Sparse Convolution hot path of hailanyi/VirConv.

  virconv_amd.csrc/            hand-written HIP kernels + the C ABI (include/virconv_hip.h -> libvirconv_hip.so)
  virconv_amd._lib             ctypes binding of the C ABI
  virconv_amd.backend_hip      torch-tensor plumbing over the ABI (device memory, current stream)
  virconv_amd.ops              rulebooks + autograd Functions
  virconv_amd.spconv           spconv-compatible operator facade (SparseConvTensor, SubMConv3d, ...)
  virconv_amd.backbone         VirConvL8x / VirConv8x / NRConvBlock (drop-in for pcdet.models.backbones_3d)
  virconv_amd.data             host-side input point discard + GPU voxeliser wrapper
  virconv_amd.parallel         one-process-per-GPU helpers (RCCL via torch.distributed)
  virconv_amd.synth            seeded synthetic KITTI-shaped scenes
"""
__version__ = "0.1.0"
