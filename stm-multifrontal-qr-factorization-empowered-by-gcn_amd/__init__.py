"""MI355X-native numerical-factorization path for STM-Multifrontal-QR.

The product is the C-ABI shared library ``libstmmqr_hip.so`` (sources in ``csrc/``, interface in
``include/stmmqr_hip.h``).  This package is the thin host-side mirror of the reference's interface for that
path (same names and argument meaning as STMMQR/include/SparseQR.h) on numpy arrays; it contains no numeric
code and no CPU fallback: importing :mod:`capi` raises if the HIP library has not been built.
"""
from .capi import (HipQR, Relax, SparseQR, analyze, relax_for_qr, QRNumeric, StmmqrError, device_alloc, device_count, device_free, device_name, get_options,  # noqa: F401
                   last_seam_ms, lib,
                   lib_path, qr_assemble, qr_cpack, qr_factorize, qr_fcsize, qr_front, qr_fsize, qr_hpinv, qr_larftb, qr_rhpack,
                   qr_stranspose2, read_matrix_market, set_options, qr_factorize_seam, plan_cache_clear, SeamNumeric, RcclTransport, CallbackTransport, device_copy)
