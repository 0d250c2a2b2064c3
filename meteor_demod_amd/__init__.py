"""meteor_demod_amd — MI355X-native LRPT demodulator (IQ in -> soft QPSK out).

Drop-in for the demod.c / dsp/ path of dbdexter-dev/meteor_demod behind a C-ABI
(include/meteor_demod_amd.h).  This package holds the HIP kernels (csrc/), the
built shared library (lib/) and a thin host-side mirror of the reference's
interface (demod.py).  Nothing here computes on the CPU.
"""
from .demod import DemodConfig, Demodulator, derive_tables, scale_freq_max  # noqa: F401

__all__ = ["DemodConfig", "Demodulator", "derive_tables", "scale_freq_max"]
