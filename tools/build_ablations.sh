#!/bin/bash
# Developer probe: side builds of liblvdhip.so for timing experiments on the 8-wave ping-pong GEMM main loops (gemm_ring.hip):
#   abl1: MFMAs compiled out (fragments kept live)      abl2: in-loop DMA compiled out      trace: s_memtime phase stamps
#   strace: s_memtime stamps at the phase and epilogue edges of the persistent walker (gemm_stream.hip)
# Results of the ablation libraries are wrong by construction; select one with LVD_LIB=<path> for timing only.
#   tools/build_ablations.sh   ->  build/abl/liblvdhip_{abl1,abl2,trace}.so
set -e
cd "$(dirname "$0")/../llm-groundedvideodiffusion_amd/csrc"
out=../../build/abl
mkdir -p $out
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result"
others=$(ls *.o | grep -v gemm_ring.o)
hipcc $FL -DLVD_ABL=1 -c gemm_ring.hip -o $out/gemm_ring_abl1.o &
hipcc $FL -DLVD_ABL=2 -c gemm_ring.hip -o $out/gemm_ring_abl2.o &
hipcc $FL -DLVD_TRACE -c gemm_ring.hip -o $out/gemm_ring_trace.o &
hipcc $FL -DLVD_TRACE -c gemm_stream.hip -o $out/gemm_stream_trace.o &
wait
for a in abl1 abl2 trace; do
  hipcc -shared -fPIC --offload-arch=gfx950 $others $out/gemm_ring_$a.o -o $out/liblvdhip_$a.so
done
# strace: s_memtime stamps in the persistent walker (gemm_stream.hip; tools/stream_trace.py)
hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v gemm_stream.o) $out/gemm_stream_trace.o -o $out/liblvdhip_strace.so
ls -la $out/*.so
