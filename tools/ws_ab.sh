#!/bin/bash
# GPU-box side of the interleaved A/B: bash tools/ws_ab.sh [mode] [rounds]
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/wsab
rocprofv3 --kernel-trace -d /tmp/wsab -o t --output-format csv -- python $REPO/tools/ws_ab.py "$@" > /tmp/wsab.log 2>&1 || tail -5 /tmp/wsab.log
python $REPO/tools/ws_ab_report.py /tmp/wsab /tmp/wsab.log
