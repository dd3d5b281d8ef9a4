#!/bin/bash
# Compile check of the HM adapter against the reference's own headers, where they lie (no reference build system is run,
# nothing is copied).  Same in-flight TypeDef.h filter as oracle/ref/build_ref.sh (MSVC-isms `#define X 1;`).
# Output: oracle/_ref/obj/TEncCuFcu.o (git-ignored).  Linking it into a full HM is the maintainer's step.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
[ -d "$REF/Lib/TLibCommon" ] || { echo "no reference tree at $REF"; exit 0; }
OUT="$HERE/../oracle/_ref/obj"; mkdir -p "$OUT"
fix() { sed -E 's/^(#define[ \t]+(GET_SATD|SKIP_RDO_ENABLE)[ \t]+)1[ \t]*;/\10/; s/^(#define[ \t]+[A-Za-z_0-9]+[ \t]+[0-9]+)[ \t]*;/\1/; s/^#endif;/#endif/' "$REF/Lib/TLibCommon/TypeDef.h"; }
g++ -std=c++11 -O2 -Wall -Wno-unused -Wno-sign-compare -Wno-reorder -Wno-parentheses -Wno-misleading-indentation -Wno-deprecated -Wno-unknown-pragmas -fpermissive -fPIC \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$REF/Lib" -I"$REF/Lib/TLibEncoder" -I"$HERE" -I"$HERE/../include" \
    -include <(fix) -include limits -c "$HERE/TEncCuFcu.cpp" -o "$OUT/TEncCuFcu.o"
echo "compiled $OUT/TEncCuFcu.o"
nm -C --defined-only "$OUT/TEncCuFcu.o" | grep " T TEncCu::" | sort
