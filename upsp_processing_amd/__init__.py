"""upsp_processing_amd -- MI355X-native engine for the ray-cast + per-frame projection
hot path of nasa/upsp-processing.

Layout
  csrc/          HIP kernels (gfx950) + C ABI (include/upsp_gpu.h) + pybind11 `raycast`
  _capi.py       ctypes view of the C ABI (fails loudly when libupsp_gpu.so is missing)
  engine.py      device-resident wrappers (torch tensors carry the HBM buffers)
  visibility.py  mirror of upsp.cam_cal_utils.visibility.VisibilityChecker
  psp.py         phase-1 driver: projection build + frame loop + reductions (psp_process)
  distributed.py frame sharding across GPUs + RCCL exchange of the time series
  synthetic.py   deterministic synthetic meshes / cameras / frames
"""
__version__ = "0.1.0"
