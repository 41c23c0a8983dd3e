#!/bin/bash
# Timing-probe builds of the library (WRONG results, same instruction streams otherwise) for upper bounds on what a part of the
# convolution kernels costs:  libsd_probe_nodma.so = no LDS-DMA instruction is issued (halo / weight fetches cost nothing),
# libsd_probe_nobar.so = no s_barrier in k_conv_mfma (waves of a workgroup run free).  A/B: tools/ab_layers.py <arch> <act> <libs...>
set -e
REPO=$(cd $(dirname $0)/.. && pwd)
for P in ${PROBES:-nodma nobar}; do
  T=/tmp/sd_probe_$P; rm -rf $T; mkdir -p $T/syconn_amd; cp -r $REPO/include $T/; cp -r $REPO/syconn_amd/csrc $T/syconn_amd/; rm -f $T/syconn_amd/csrc/*.o
  EXTRA_FLAGS=
  if [ $P = nodma ]; then
    python3 - $T/syconn_amd/csrc/sd_device.h <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
a = s.index('__device__ __forceinline__ void glds16(')
b = s.index('}', a)
s = s[:a] + '__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) { (void)g; (void)lds_wave_base; }' + s[b + 1:]
open(p, 'w').write(s)
# ... and the halo / weight slots are filled ONCE with pseudo-random 16-bit floats of magnitude 0.06 ... 0.5 and random sign: with LDS
# left at zero the matrix cores multiply zeros, the chip draws less power and clocks ~19 % higher (MI355X_MICROARCH.md, "DVFS
# give-back") -- the probe would then measure the clock, not the DMA
p = sys.argv[1].replace('sd_device.h', 'sd_conv_mfma.h'); s = open(p).read()
anchor = '    // ordinary (VGPR-destination) global loads happen only here, before the first DMA is issued\n'
assert anchor in s
s = s.replace(anchor, anchor + '''    for (unsigned i = threadIdx.x; i < (unsigned)(reinterpret_cast<char*>(wl) - smem) / 4; i += WAVES * 64) {
        unsigned h = i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        reinterpret_cast<unsigned*>(smem)[i] = (0x3d803d80u + ((h & 0x17f) | ((h << 7) & 0x017f0000u))) | ((h >> 3) & 0x80008000u);
    }
    __syncthreads();
''', 1)
open(p, 'w').write(s)
PY
  elif [ ${P#dec0_} != $P ]; then
    X=$(echo ${P#dec0_} | tr a-z A-Z)      # dec0_no_final -> -DSD_PROBE_DEC0_NO_FINAL etc. (k_dec0 without one of its pieces)
    EXTRA_FLAGS="-DSD_PROBE_DEC0_$X"
  else
    sed -i 's/\\n\\ts_barrier//g; s/asm volatile("s_barrier" ::: "memory");/asm volatile("s_nop 0" ::: "memory");/g' $T/syconn_amd/csrc/sd_conv_mfma.h
  fi
  make -C $T/syconn_amd/csrc -j4 EXTRA="${EXTRA_FLAGS:-}" > $T/build.log 2>&1; EXTRA_FLAGS=
  cp $T/syconn_amd/libsyconn_dense_hip.so $REPO/syconn_amd/libsd_probe_$P.so
  echo built $REPO/syconn_amd/libsd_probe_$P.so
done
