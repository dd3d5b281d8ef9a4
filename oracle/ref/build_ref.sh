#!/bin/bash
# ORACLE / test infrastructure.  Builds oracle/_ref/libhmleaf.so: the reference's own LEAF code
# (tables, intra prediction, reference samples, SATD/SSE, transforms, RDOQ, dequantiser, CABAC
# bit counter and residual syntax) compiled from the sources WHERE THEY LIE under /root/reference.
#
# * No reference source is copied or written anywhere: the only change the sources need to get
#   through GCC -- `#define NAME <int>;` / `#endif;` in TLibCommon/TypeDef.h:55-131 (MSVC-isms) --
#   is applied in flight by feeding a sed-filtered TypeDef.h through `-include /dev/fd/N`; its
#   include guard then skips the on-disk copy.
# * The CU/PU search loops (TEncCu.cpp, TEncSearch.cpp, TEncTop.cpp) include OpenCV headers that
#   this image lacks; they are NOT built (no stand-in headers), see oracle/README.md.
# * Output goes only to oracle/_ref/ (git-ignored).
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/../_ref"
[ -d "$REF/Lib/TLibCommon" ] || { echo "no reference tree at $REF: keeping prebuilt $OUT"; exit 0; }
mkdir -p "$OUT/obj"
COMMON="TComRom TComTrQuant TComRdCost TComRdCostWeightPrediction TComPrediction TComPattern TComInterpolationFilter \
TComWeightPrediction TComYuv TComPicYuv TComPicYuvMD5 TComPic TComPicSym TComDataCU TComSlice TComTU TComChromaFormat TComMotionInfo \
TComBitStream ContextModel ContextModel3DBuffer TComCABACTables Debug SEI TComSampleAdaptiveOffset TComLoopFilter"
ENC="TEncSbac TEncEntropy TEncBinCoderCABAC TEncBinCoderCABACCounter"
FLAGS="-std=c++11 -O2 -w -fPIC -fpermissive -ffp-contract=off -I$REF/Lib"
fix() { sed -E 's/^(#define[ \t]+[A-Za-z_0-9]+[ \t]+[0-9]+)[ \t]*;/\1/; s/^#endif;/#endif/' "$REF/Lib/TLibCommon/TypeDef.h"; }
objs=""
for f in $COMMON; do
  o="$OUT/obj/$f.o"; s="$REF/Lib/TLibCommon/$f.cpp"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ]; then g++ $FLAGS -include <(fix) -c "$s" -o "$o"; fi
  objs="$objs $o"
done
for f in $ENC; do
  o="$OUT/obj/$f.o"; s="$REF/Lib/TLibEncoder/$f.cpp"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ]; then g++ $FLAGS -include <(fix) -c "$s" -o "$o"; fi
  objs="$objs $o"
done
g++ $FLAGS -include <(fix) -c "$REF/Lib/libmd5/libmd5.c" -x c -o "$OUT/obj/libmd5.o" 2>/dev/null || gcc -O2 -w -fPIC -c "$REF/Lib/libmd5/libmd5.c" -o "$OUT/obj/libmd5.o"
g++ $FLAGS -include <(fix) -I"$HERE" -c "$HERE/ref_driver.cpp" -o "$OUT/obj/ref_driver.o"
g++ -shared -o "$OUT/libhmleaf.so" $objs "$OUT/obj/libmd5.o" "$OUT/obj/ref_driver.o"
rm -rf "$OUT/obj"
echo "built $OUT/libhmleaf.so"
