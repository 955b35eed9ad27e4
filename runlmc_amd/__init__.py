"""runlmc_amd -- MI355X-native matrix-free LMC inference hot path.

Drop-in for the hot path of vlad17/runlmc (runlmc.linalg / runlmc.approx /
runlmc.lmc): the structured-operator products, the batched Krylov solves and
the Hutchinson gradient run as hand-written HIP kernels (gfx950) behind the C
ABI in include/runlmc_hip.h.  There is no CPU fallback: constructing an
operator without the native library (or without a GPU) raises.
"""
__version__ = '0.1.0'
