#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python3 profiles/predict_probe.py 2>&1 | tail -8
