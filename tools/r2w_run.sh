#!/bin/bash
# round-2 third session: sanity after the container rebuild + half-batch PMC + the default bench line with also.c5.filter_batch
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2w; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu_tail.txt
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_default_line.json
timeout 900 bash tools/tools_profile_batch_pmc.sh f16 384 4 $O/pmc_f16_dot > $O/half_batch_pmc_f16_dot.txt 2>&1
cat $O/smoke.txt | tail -2; cat $O/pytest_gpu_tail.txt; cut -c1-400 $O/bench_default_line.json; cat $O/half_batch_pmc_f16_dot.txt
