#!/bin/bash
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 profiles/probes/stream_hop.hip -o /tmp/stream_hop && timeout -k 5 120 /tmp/stream_hop
