#!/bin/bash
# round 4, thirty-first GPU call: which counters exist for address translation and for where fabric reads are served (list only)
export TMPDIR=/tmp
rocprofv3 --list-avail > gpurun_out/r04_counters_avail.txt 2>&1; echo "rc=$?"
grep -c . gpurun_out/r04_counters_avail.txt
grep -i -E "UTCL|TLB|TRANSLATION|MALL|DRAM|EA0_RDREQ|TCP_TCC|TCP_PENDING|TCP_TA|TA_BUSY|TCP_GATE|TCP_READ" gpurun_out/r04_counters_avail.txt | cut -c1-200 | sort -u | head -80
