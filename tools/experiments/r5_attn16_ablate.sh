# timing-only ablations of lg_attn16_kernel (libraries built from a patched copy: -DABL_NOEXP / NOC / NOA); device time of the
# single-pair SP+LightGlue graph replay, 18 attention launches per forward
python -m pytest tests -m gpu -x -q -k "lightglue or lg or small_grid" > gpurun_out/s2_attn16.log 2>&1; tail -1 gpurun_out/s2_attn16.log
echo base; python tools/latency_graph.py 2>&1 | tail -1
for v in NOEXP NOC NOA; do echo $v; EINX_LIB=ab_libs/libeinx_$v.so EINX_ALLOW_TIMING_ONLY=1 python tools/latency_graph.py 2>&1 | tail -1; done
