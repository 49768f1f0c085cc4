#!/bin/bash
# round 4 batch p3: FL_OPT_MGCG_FUSE A/B inside one process (tools/mgcg_fuse_ab.py), twice
set -o pipefail
O=gpurun_out/r04p3; mkdir -p $O
timeout -k 10 500 python3 tools/mgcg_fuse_ab.py > $O/ab1.json 2>$O/ab1.err; echo "ab1 rc=$?"; cat $O/ab1.json; tail -3 $O/ab1.err
timeout -k 10 500 python3 tools/mgcg_fuse_ab.py > $O/ab2.json 2>$O/ab2.err; echo "ab2 rc=$?"; cat $O/ab2.json
