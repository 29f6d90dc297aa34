#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python3 profiles/scales_probe.py 2>&1 | tail -3
