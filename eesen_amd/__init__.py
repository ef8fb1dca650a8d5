"""eesen_amd: an MI355X-native (gfx950) CTC training path behind Eesen's net/netbin API.

Only the hot path of `train-ctc-parallel` lives here: HIP kernels + a C-ABI library in
`csrc/` and the Python mirror of the reference's `Net` / `Ctc` operator interface in `api.py`.
"""
__version__ = "0.1.0"
