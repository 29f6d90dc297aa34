"""Random-row gather rate against workgroups per CU (requests in flight): fmx_measure_gather_occ pads every workgroup of the probe
with dynamic LDS so that at most 160 KiB / lds workgroups share a CU.  256 threads per workgroup, 4 lanes per 64-byte row,
4 or 8 rows in flight per lane."""
import ctypes as C, sys
sys.path.insert(0, ".")
from fmwr_amd import _lib as L
for table_mb in (16.8, 64.0):
    for infl in (4, 8):
        row = []
        for lds in (0, 20480, 32768, 40960, 65536):
            out = C.c_double()
            L.check(L.lib().fmx_measure_gather_occ(0, C.c_int64(int(table_mb * 1e6) // 64 * 64), 64, C.c_int64(262144), 32, infl, 20, lds, C.byref(out)))
            row.append(f"{'max' if lds == 0 else 163840 // lds} wg/CU: {out.value / 1e9:5.1f}")
        print(f"table {table_mb:5.1f} MB, {infl} in flight per lane | " + " | ".join(row) + "  G rows/s")
