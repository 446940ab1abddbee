#!/bin/bash
# Round evidence, collected on the GPU box (gpurun, from the repo root):
#   bench lines of the five reported configurations, rocprofv3 kernel-trace/stats summaries
#   (overlapped, single-stream, LightGlue) and the --pmc passes over the dominant kernel.
# Results land in gpurun_out/ev/; tools/assemble_profiles.py turns them into profiles/rNN_*.
# Usage: tools/collect_profiles.sh [bench|traces|pmc|all]   (a gpurun call is limited to 20 minutes: run the phases separately)
set -e
PHASE=${1:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev
mkdir -p $O
cd $R
if [ $PHASE = bench ] || [ $PHASE = all ]; then
python bench.py > $O/bench_sp_mnn.json 2> $O/bench_sp_mnn.err
python bench.py --log-assignment --dense --no-cpu-baseline --no-extras > $O/bench_sp_mnn_full.json 2>> $O/bench.err
python bench.py --with-metrics --no-cpu-baseline --no-extras > $O/bench_sp_mnn_metrics.json 2>> $O/bench.err
python bench.py --config sp_lg --cpu-pairs 2 > $O/bench_sp_lg.json 2>> $O/bench.err
python bench.py --config silk_mnn --cpu-pairs 2 > $O/bench_silk.json 2>> $O/bench.err
python bench.py --config silk_lg --cpu-pairs 1 --no-extras > $O/bench_silk_lg.json 2>> $O/bench.err
# one-rank RCCL group through bench.py's own launcher, and through torchrun (the driver's launch pattern)
python bench.py --gpus 1 --spawn --no-cpu-baseline --no-extras > $O/bench_spawn1.json 2> $O/bench_spawn1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-extras > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
python bench.py --layer-table > $O/layer_table.txt 2>> $O/bench.err
python tools/latency_b1.py 1 > $O/latency_b1.txt 2>> $O/bench.err
python tools/latency_graph.py > $O/latency_graph.txt 2>> $O/bench.err
python tools/events_bench.py > $O/events_bench.txt 2>> $O/bench.err
echo "bench lines done"
fi
cd /tmp && export TMPDIR=/tmp
if [ $PHASE = traces ] || [ $PHASE = all ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_overlap -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-scale-legs > $O/prof_overlap.log 2>&1
EINX_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-scale-legs > $O/prof_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lg -o p -- python3 $R/bench.py --config sp_lg --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_lg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kernel_only -o p -- python3 $R/bench.py --kernel-only > $O/prof_kernel_only.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dense -o p -- python3 $R/tools/up_bench.py > $O/prof_dense.log 2>&1
python3 $R/tools/up_bench.py --ref > $O/up_bench.txt 2>/dev/null
# single pairs: kernel list of 620 forwards each (SP+MNN, SP+LightGlue)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1_mnn -o p -- python3 $R/tools/latency_b1.py 1 SP_MNN > $O/prof_b1_mnn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1_lg -o p -- python3 $R/tools/latency_b1.py 1 SP_LG > $O/prof_b1_lg.log 2>&1
python3 $R/tools/latency_b1.py 1 SP_LG > $O/latency_b1_lg.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_events -o p -- python3 $R/tools/events_bench.py > $O/prof_events.log 2>&1
echo "kernel traces done"
fi
if [ $PHASE = pmc ] || [ $PHASE = all ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --kernel-only > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_busy -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-scale-legs > $O/pmc_busy.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_busy_lg -o p -- python3 $R/tools/lg_bench.py --skip-linear --reps 1 > $O/pmc_busy_lg.log 2>&1
echo "pmc done"
fi
ls $O
