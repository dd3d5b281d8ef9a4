#!/bin/bash
# ORACLE / test infrastructure.  Builds oracle/_ref/libhmleaf.so: the reference's own code for the hot path
# compiled from the sources WHERE THEY LIE under /root/reference:
#   * leaves: tables, intra prediction, reference samples, SATD/SSE/SAD, transforms, RDOQ, dequantiser, CABAC bit
#     counter and syntax writers, TComDataCU (MPM / merge / AMVP derivation), interpolation filter, loop filter, SAO;
#   * TLibEncoder/TEncSearch.cpp -- the PU/TU SEARCH LOOPS (estIntraPredLumaQT / ChromaQT, xRecurIntraCodingLumaQT,
#     predInterSearch, xMotionEstimation, xPatternSearch, xPatternSearchFracDIF, encodeResAndCalcRdInterCU,
#     xEstimateInterResidualQT ...);
#   * the fork's TCM threshold fit, four free functions of TLibEncoder/TEncSlice.cpp (:193-392).
#
# No reference source is copied or written anywhere and no stand-in header / library is provided.  What the sources
# need to get through GCC is applied IN FLIGHT:
#   1. `#define NAME <int>;` / `#endif;` in TLibCommon/TypeDef.h:55-131 (MSVC-isms): a sed-filtered TypeDef.h is fed
#      through `-include /dev/fd/N`; its include guard then skips the on-disk copy.
#   2. The same filter sets two of the fork's own feature switches to 0 -- GET_SATD and SKIP_RDO_ENABLE
#      (TypeDef.h:74-75).  They guard the only blocks of TEncSearch.cpp that call into tools_YS.cpp (which needs
#      OpenCV): the SATD bookkeeping for the feature dump and the SVM `SkipRDO` early-out, which the fork's default
#      control (Naive model, tools_YS.cpp:4-58) never enables.  Every class layout (TComDataCU, TComPic) keeps the
#      fork's members; all translation units see the same switches.
#   3. TEncSearch.cpp:43-44 `#include "TLibCommon/linear.h"` / `"TLibCommon/tools_YS.h"` (-> cvheaders.h -> OpenCV,
#      absent in this image) are dropped in flight (the file is piped through sed into g++); with switch 2 nothing in
#      the translation unit refers to them.
#   4. TEncSlice.cpp needs `-include limits`.  Only its four free TCM functions are kept: everything is compiled with
#      -ffunction-sections and linked with --gc-sections under a version script that exports ref_* and those four
#      symbols, so the TEncSlice member functions (which reference TEncTop / TEncGOP, not built) are discarded.
# TEncCu.cpp (xCompressCU itself) and tools_YS.cpp still cannot be built (their fork code is not behind switches) and
# are NOT built; see oracle/README.md.
# Output goes only to oracle/_ref/ (git-ignored).
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/../_ref"
[ -d "$REF/Lib/TLibCommon" ] || { echo "no reference tree at $REF: keeping prebuilt $OUT"; exit 0; }
mkdir -p "$OUT/obj"
COMMON="TComRom TComTrQuant TComRdCost TComRdCostWeightPrediction TComPrediction TComPattern TComInterpolationFilter \
TComWeightPrediction TComYuv TComPicYuv TComPicYuvMD5 TComPic TComPicSym TComDataCU TComSlice TComTU TComChromaFormat TComMotionInfo \
TComBitStream ContextModel ContextModel3DBuffer TComCABACTables Debug SEI TComSampleAdaptiveOffset TComLoopFilter"
ENC="TEncSbac TEncEntropy TEncBinCoderCABAC TEncBinCoderCABACCounter TEncSampleAdaptiveOffset"
FLAGS="-std=c++11 -O2 -w -fPIC -fpermissive -ffp-contract=off -ffunction-sections -fdata-sections -I$REF/Lib -I$REF/Lib/TLibEncoder"
fix() { sed -E 's/^(#define[ \t]+(GET_SATD|SKIP_RDO_ENABLE)[ \t]+)1[ \t]*;/\10/; s/^(#define[ \t]+[A-Za-z_0-9]+[ \t]+[0-9]+)[ \t]*;/\1/; s/^#endif;/#endif/' "$REF/Lib/TLibCommon/TypeDef.h"; }
stamp="$OUT/obj/.flags"; echo "$FLAGS $(fix | md5sum)" > "$stamp.new"
if ! cmp -s "$stamp.new" "$stamp"; then rm -f "$OUT"/obj/*.o; mv "$stamp.new" "$stamp"; else rm -f "$stamp.new"; fi
objs=""; pids=""
cc() { # $1 = object, $2 = source
  if [ ! -f "$1" ] || [ "$2" -nt "$1" ]; then g++ $FLAGS -include <(fix) -include limits -c "$2" -o "$1" & pids="$pids $!"; fi; objs="$objs $1"; }
for f in $COMMON; do cc "$OUT/obj/$f.o" "$REF/Lib/TLibCommon/$f.cpp"; done
for f in $ENC TEncSlice; do cc "$OUT/obj/$f.o" "$REF/Lib/TLibEncoder/$f.cpp"; done
o="$OUT/obj/TEncSearch.o"; s="$REF/Lib/TLibEncoder/TEncSearch.cpp"
if [ ! -f "$o" ] || [ "$s" -nt "$o" ]; then
  sed -E '/^#include "TLibCommon\/(linear|tools_YS)\.h"/d' "$s" | g++ $FLAGS -include <(fix) -include limits -x c++ -c - -o "$o" & pids="$pids $!"
fi
objs="$objs $o"
# TEncCu.cpp without the body of xCompressCU (lines 456-1616: its fork code calls into tools_YS.cpp / OpenCV and is not behind a
# switch), without the fork's two #includes and two extern model pointers (:48-49, :56-57), without xCheckRDCostIntra_Rough
# (:78-156, dead with SKIP_RDO_ENABLE 0) and without the g_iQP bookkeeping line (:1650).  What remains is the reference's own
# xCheckRDCostMerge2Nx2N, xCheckRDCostInter, xCheckRDCostIntra, xCheckBestMode, deriveTestModeAMP, xCopyYuv2Pic/Tmp ...
# The class is renamed in flight (-DTEncCu=TEncCuRef): the repository's adapter defines TEncCu's public methods in this library.
o="$OUT/obj/TEncCuRef.o"; s="$REF/Lib/TLibEncoder/TEncCu.cpp"
if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$0" -nt "$o" ]; then
  sed -E '48,49d;56,57d;78,156d;456,1616d;1650d' "$s" | g++ $FLAGS -DTEncCu=TEncCuRef -include <(fix) -include limits -x c++ -c - -o "$o" & pids="$pids $!"
fi
objs="$objs $o"
for p in $pids; do wait $p; done
# the HM adapter of this repository (adapter/TEncCuFcu.cpp: TEncCu's public methods over libfcu.so) is compiled against the
# same headers and linked in, so that tests can push fcu_ctu_out data through its marshalling into a real TComDataCU and
# through its encodeCtu walk into the reference's own TEncEntropy (ref_adapter_* in ref_driver.cpp)
PKG="$HERE/../../fast-cu-decision-hevc_amd"
ADP="$HERE/../../adapter"
a="$OUT/obj/TEncCuFcu.o"
if [ ! -f "$a" ] || [ "$ADP/TEncCuFcu.cpp" -nt "$a" ] || [ "$ADP/fcu_marshal.h" -nt "$a" ] || [ "$HERE/../../include/fcu.h" -nt "$a" ]; then
  g++ $FLAGS -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$ADP" -I"$HERE/../../include" -include <(fix) -include limits -c "$ADP/TEncCuFcu.cpp" -o "$a"
fi
objs="$objs $a"
gcc -O2 -w -fPIC -c "$REF/Lib/libmd5/libmd5.c" -o "$OUT/obj/libmd5.o"
g++ $FLAGS -include <(fix) -include limits -I"$HERE" -I"$ADP" -I"$HERE/../../include" -c "$HERE/ref_driver.cpp" -o "$OUT/obj/ref_driver.o"
cat > "$OUT/obj/export.map" <<'MAP'
{ global: ref_*; fcu_adapter_last_record; _Z21TCMprocessOneSequencePiiS_PdS0_S0_; _Z14FindStartPointP7tBucketii; _Z17ComputeLikelyhoodiiP7tBucketi; _Z20ComputeLambdaGivenYcddd; local: *; };
MAP
g++ -shared -o "$OUT/libhmleaf.so" $objs "$OUT/obj/libmd5.o" "$OUT/obj/ref_driver.o" -Wl,--gc-sections -Wl,--version-script="$OUT/obj/export.map" -Wl,-z,defs \
    -L"$PKG" -lfcu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../../fast-cu-decision-hevc_amd' -Wl,-rpath,/opt/rocm/lib
echo "built $OUT/libhmleaf.so"
