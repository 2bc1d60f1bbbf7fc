#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_INST_LDS\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_LDS_[A-Z_]*" | sort -u > $OUT/avail_sq.txt; wc -l $OUT/avail_sq.txt; tr '\n' ' ' < $OUT/avail_sq.txt
