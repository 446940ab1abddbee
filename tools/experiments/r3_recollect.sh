set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev
mkdir -p $O; cd /tmp && export TMPDIR=/tmp

rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_overlap -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-scale-legs > $O/prof_overlap.log 2>&1
EINX_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-scale-legs > $O/prof_single.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_busy -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-scale-legs > $O/pmc_busy.log 2>&1
echo done
