#!/bin/bash
# tools/make_probe_libs.sh -- builds the timing-only PROBE variants of libcvsteer_hip.so that tools/ab_build.py compares
# (results in profiles/r02_probe_builds.txt).  Each variant is the product kernel source with one line changed by sed;
# results of these builds are WRONG on purpose (stores or loads dropped by the hardware range check, ...) -- they only
# price a part of the kernel.  Output: tools/ablibs/<name>.so (git-ignored).  No GPU needed to build.
#   cur      the product library as it stands
#   NOSTORE  every store of the basis kernels dropped (lane offset out of range)      -> compute + loads only
#   NOLOAD   every load dropped (returns 0)                                          -> the filters run on zeros
#   NOMAIN / NOHALO   only the 64-lane row load / only the 2W-lane halo load dropped
#   HOT      every load reads the first 64 KB of rows (always a cache hit)
#   AUX<n>   streaming stores issued with cache-policy bits n instead of 2 (nt): 3 = nt sc0, 18 = nt sc1, 19 = nt sc0 sc1, 16 = sc1, 17 = sc0 sc1
#   XVALU / XSALU   12 extra independent v_mov / s_mov per row and wave
#   PW1     pipeline epilogue with ONE cos/sin evaluation for the three phase weights and none for 2*theta (prices those evaluations)
#   LDAUX<n> input loads issued with cache-policy bits n (2 = nt, 1 = sc0, 3 = nt sc0, 16 = sc1, 18 = nt sc1)
#   WPB8    eight waves (512 columns) per workgroup instead of four; WPB1 / WPB2: one / two
#   PAIRW<n> the G4 pair kernel compiled for n waves per SIMD (register budget 512/n)
#   ILP     same source, machine scheduler strategy max-ilp (independent accumulation chains interleaved)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/cvsteer_amd/csrc
O=$R/tools/ablibs
mkdir -p $O
make -C $C -s
cp $R/cvsteer_amd/libcvsteer_hip.so $O/cur.so
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -I$R/include -I$C -ffp-contract=off -fno-slp-vectorize"
build() {  # name, sed program (or python for multi-line) [, extra compiler flags]
  name=$1; shift
  sed -e "$1" $C/cvs_kernels_basis.hip > $C/_probe_$name.hip
  /opt/rocm/bin/hipcc $FLAGS ${2:-} -c $C/_probe_$name.hip -o /tmp/basis_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $O/$name.so $C/cvs_api.o $C/cvs_batch.o $C/cvs_state.o $C/cvs_taps.o /tmp/basis_$name.o $C/cvs_kernels_point.o -ldl -lpthread
  rm -f $C/_probe_$name.hip
  echo "built $O/$name.so"
}
for v in "$@"; do
  case $v in
    NOSTORE) build $v 's/__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, lane_off,/__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, 0x80000000u,/' ;;
    NOLOAD)  build $v 's/__builtin_amdgcn_raw_buffer_load_b32(r, lane_off,/__builtin_amdgcn_raw_buffer_load_b32(r, 0x80000000u,/' ;;
    NOMAIN)  build $v 's/pre\[j\] = bld(r_in, nxmb, ro);/pre[j] = bld(r_in, kLaneOff, ro);/; s/pre\[j\] = bld(r_in, xmb, ro);/pre[j] = bld(r_in, kLaneOff, ro);/' ;;
    NOHALO)  build $v 's/preh\[j\] = bld(r_in, nxhb, ro);/preh[j] = bld(r_in, kLaneOff, ro);/; s/preh\[j\] = bld(r_in, xhb, ro);/preh[j] = bld(r_in, kLaneOff, ro);/' ;;
    HOT)     build $v 's/__builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0)/__builtin_amdgcn_raw_buffer_load_b32(r, lane_off, (row_off) \& 0xffffu, 0)/' ;;
    AUX*)    build $v "s/r, lane_off, row_off, STREAM ? 2 : 0);/r, lane_off, row_off, STREAM ? ${v#AUX} : 0);/" ;;
    XVALU)   build $v 's|^            // ---- column pass on the window; newest row is slot j, centre is W rows back ----$|            { float dmy; asm volatile("v_mov_b32 %0, 1.0\\nv_mov_b32 %0, 2.0\\nv_mov_b32 %0, 1.0\\nv_mov_b32 %0, 2.0\\nv_mov_b32 %0, 1.0\\nv_mov_b32 %0, 2.0\\nv_mov_b32 %0, 1.0\\nv_mov_b32 %0, 2.0\\nv_mov_b32 %0, 1.0\\nv_mov_b32 %0, 2.0\\nv_mov_b32 %0, 1.0\\nv_mov_b32 %0, 2.0" : "=v"(dmy)); }|' ;;
    XSALU)   build $v 's|^            // ---- column pass on the window; newest row is slot j, centre is W rows back ----$|            { int dmy; asm volatile("s_mov_b32 %0, 1\\ns_mov_b32 %0, 2\\ns_mov_b32 %0, 1\\ns_mov_b32 %0, 2\\ns_mov_b32 %0, 1\\ns_mov_b32 %0, 2\\ns_mov_b32 %0, 1\\ns_mov_b32 %0, 2\\ns_mov_b32 %0, 1\\ns_mov_b32 %0, 2\\ns_mov_b32 %0, 1\\ns_mov_b32 %0, 2" : "=s"(dmy)); }|' ;;
    ILP)     build $v 's/^$//' "-mllvm -amdgpu-sched-strategy=max-ilp" ;;
    PW1)     build $v 's/q\[6\] = __fmul_rn(en, phase_lambda<true>(q\[4\], 0.f, true));/q[6] = __fmul_rn(en, q[5]);/; s/q\[7\] = __fmul_rn(en, phase_lambda<true>(q\[4\], kPiF, true));/q[7] = __fmul_rn(q[5], q[6]);/; s/sincos_small(__fmul_rn(th, 2.0f), s2, cc2);/s2 = th * c2; cc2 = th * c3;/' ;;
    LDAUX*)  build $v "s/__builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0)/__builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, ${v#LDAUX})/" ;;
    WPB1)    build $v 's/constexpr int wpb = 4;/constexpr int wpb = 1;/; s/#define CVS_LAUNCH_B(FL, BATCHED) CVS_LAUNCH_W(FL, BATCHED, 4)/#define CVS_LAUNCH_B(FL, BATCHED) CVS_LAUNCH_W(FL, BATCHED, 1)/' ;;
    WPB2)    build $v 's/constexpr int wpb = 4;/constexpr int wpb = 2;/; s/#define CVS_LAUNCH_B(FL, BATCHED) CVS_LAUNCH_W(FL, BATCHED, 4)/#define CVS_LAUNCH_B(FL, BATCHED) CVS_LAUNCH_W(FL, BATCHED, 2)/' ;;
    WPB8)    build $v 's/constexpr int wpb = 4;/constexpr int wpb = 8;/; s/#define CVS_LAUNCH_B(FL, BATCHED) CVS_LAUNCH_W(FL, BATCHED, 4)/#define CVS_LAUNCH_B(FL, BATCHED) CVS_LAUNCH_W(FL, BATCHED, 8)/' ;;
    PAIRW*)  build $v "s/__global__ __launch_bounds__(256) void k_basis_pair/__global__ __launch_bounds__(256, ${v#PAIRW}) void k_basis_pair/" ;;
    *) echo "unknown probe $v"; exit 2 ;;
  esac
done
