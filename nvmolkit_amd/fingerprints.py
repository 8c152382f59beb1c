"""Fingerprint packing helpers and the Morgan fingerprint generator
(same API as the reference's nvmolkit/fingerprints.py).

Bit order (defines the whole similarity path): bit ``j`` of a fingerprint is bit ``j % 32`` of
int32 word ``j // 32`` (reference: nvmolkit/fingerprints.py:25-72).
"""

from __future__ import annotations

import torch


def unpack_fingerprint(fp: torch.Tensor) -> torch.Tensor:
    """(n, fp_size/32) int32/uint32 words -> (n, fp_size) bool."""
    if fp.dtype not in (torch.int32, torch.uint32):
        raise ValueError("Input tensor must have dtype int32 or uint32")
    words = fp.view(torch.int32) if fp.dtype == torch.uint32 else fp
    shifts = torch.arange(32, device=words.device, dtype=torch.int32)
    bits = (words.unsqueeze(-1) >> shifts) & 1
    return bits.to(torch.bool).reshape(words.shape[0], words.shape[1] * 32)


def pack_fingerprint(fp: torch.Tensor) -> torch.Tensor:
    """(n, fp_size) bool -> (n, ceil(fp_size/32)) int32 words (zero padded)."""
    n, nbits = fp.shape
    nwords = (nbits + 31) // 32
    bits = fp.to(torch.bool)
    if nbits != nwords * 32:
        bits = torch.nn.functional.pad(bits, (0, nwords * 32 - nbits))
    weights = torch.ones(32, dtype=torch.int32, device=fp.device) << torch.arange(32, dtype=torch.int32,
                                                                                   device=fp.device)
    return (bits.reshape(n, nwords, 32).to(torch.int32) * weights).sum(dim=2, dtype=torch.int32)
