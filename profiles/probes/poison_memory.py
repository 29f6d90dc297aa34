"""Fill (almost) all free device memory with a byte pattern and exit: the next process's fresh allocations then land on pages that hold garbage, as on a box another tenant has just
left.  A kernel that reads memory it never wrote (a sums buffer assumed zero, a count array, a flag) shows up as a wrong result in the run that follows.
usage: python profiles/probes/poison_memory.py [byte, default 0xFF: NaN as fp32 / fp64, 4 294 967 295 as an index]"""
import sys
import torch
byte = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0xFF
free, total = torch.cuda.mem_get_info()
chunks, got = [], 0
while True:
    free, _ = torch.cuda.mem_get_info()
    n = min(free - (2 << 30), 8 << 30)
    if n < (1 << 30):
        break
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    t.fill_(byte)
    chunks.append(t); got += n
torch.cuda.synchronize()
print(f"poisoned {got / 2**30:.0f} GiB of {total / 2**30:.0f} GiB with 0x{byte:02X}")
