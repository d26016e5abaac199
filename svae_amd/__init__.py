"""svae_amd -- MI355X-native structured E-step for SVAEs (the hot path of mattjj/svae).

Host side mirrors the reference's module layout for the path it replaces:

    svae_amd.lds.lds_inference   <->  svae/lds/lds_inference.py   (E-step wrappers)
    svae_amd.models.gmm          <->  svae/models/gmm.py          (local_meanfield, run_inference)
    svae_amd.models.lds          <->  svae/models/lds.py          (run_inference glue)
    svae_amd.distributions.*     <->  svae/distributions/*.py     (global -> local maps)

All message passing runs in the HIP library ``libsvae_hip.so`` (C ABI: include/svae_hip.h) built
in-tree by ``__graft_entry__.build()`` / ``make -C svae_amd/csrc``.  There is NO CPU fallback:
importing the library wrappers without the built extension raises.
"""
__version__ = "0.1.0"
