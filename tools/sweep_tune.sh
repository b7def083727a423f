for lib in "" tools/ablate/sw_nt.so tools/ablate/sw_u2.so tools/ablate/sw_u4.so tools/ablate/sw_ntu2.so; do
  for g in 1024 2048 4096 8192 20000; do
    echo "lib=$lib grid=$g: $(TLSQ_LIB=$lib TLSQ_SWEEP_GRID=$g python tools/kbench.py sweeps --reps 30 2>&1 | grep fused)"
  done
done
