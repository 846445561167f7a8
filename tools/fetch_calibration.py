#!/usr/bin/env python3
"""What FETCH_SIZE counts for the access shapes of the encode kernels (encode_string_1p's traffic was argued
two ways).  Three reads of KNOWN size, each the only work of its launch, to be run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/fetch_calibration.py
and summarised by tools/fetch_calibration_summary.py (kernel name -> counter):

  wide      every byte of a 1 GiB buffer, 16 bytes per lane (torch's vectorised sum over int32)         -> bytes = N
  strided8  the first 8 bytes of every 16-byte row of a 1 GiB buffer (x.view(-1, 4)[:, :2]: the length pass of
            encode_string_1p reads dwords 0-1 of every string_t)                                         -> useful = N / 2, lines touched = N
  strided4  the first 4 bytes of every 16-byte row                                                       -> useful = N / 4, lines touched = N

FETCH_SIZE is in KB; MI355X_MICROARCH.md says wide coalesced reads show as half their bytes on gfx950."""
import json
import sys

import torch


def main():
    n = 1 << 30
    x = torch.randint(0, 1 << 20, (n // 4,), dtype=torch.int32, device="cuda")
    rows = x.view(-1, 4)
    torch.cuda.synchronize()
    out = {"bytes": n}
    for _ in range(3):
        out["wide"] = int(x.sum().item())
        out["strided8"] = int(rows[:, :2].sum().item())
        out["strided4"] = int(rows[:, 0].sum().item())
    torch.cuda.synchronize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
