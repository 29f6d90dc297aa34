#!/bin/bash
export TMPDIR=/tmp
bash profiles/regen_r03.sh
