"""epc-net_amd: MI355X-native EPC-Net hot path (HIP kernels behind a C ABI + the reference's Python operator API).

The directory name contains a hyphen like the reference's own module files (``models/epc-net.py``), so load it the
way the reference loads its models (``importlib.import_module``, train.py:81) -- or ``import epcnet_amd`` (repo root
alias).  Sub-modules:

  lib         ctypes binding of libepcnet_hip.so (include/epcnet.h); raises if the library is not built
  engine      inference engine (packed weights + workspace + epc_net_forward)
  variables   variable store / scopes / initialisers with the reference's checkpoint names
  tf_bundle   TensorFlow checkpoint (.index/.data) reader, no TensorFlow needed
  models/     epc-net.py, epc-net-l.py : placeholder_inputs / forward / losses  (reference models/*.py)
  loupe       PoolingBaseModel / NetVLAD / G_VLAD                                 (reference loupe.py)
  utils/      tf_util.py op wrappers                                              (reference utils/tf_util.py)
  retrieval   get_latent_vectors / get_recall                                     (reference evaluate.py)
  distributed one-process-per-GPU sharding + RCCL all-gather of descriptors
"""
__version__ = "0.1.0"
