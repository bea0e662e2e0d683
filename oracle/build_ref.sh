#!/bin/bash
# Build the checker library from the reference's own sources (never copied into this
# repository): oracle/_ref/libltref_<W>x<H>_<tables>.so
#
#   oracle/build_ref.sh <W> <H> [trans|transbs]
#
# The screen size is a compile-time constant in the reference (LightTrack.h:44-45), so
# there is one library per resolution.  <tables> picks the file that provides the T_*
# functions: Trans.cpp (correct sin/cos tables; default) or TransBs.cpp (the file the
# reference's project actually builds, whose T_init_math truncates before scaling,
# TransBs.cpp:70-71).  The sources are compiled where they lie: a scratch directory of
# symlinks lets their quote-includes of "LightTrack.h" resolve to oracle/shadow/LightTrack.h
# instead of the Win32 header.  Needs /root/reference; the outputs travel, the sources don't.
set -euo pipefail
W=${1:?width}; H=${2:?height}; TAB=${3:-trans}
REF=${LT_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF" ] || { echo "reference tree $REF not present: nothing built" >&2; exit 2; }
case "$TAB" in trans) TFILE=Trans.cpp;; transbs) TFILE=TransBs.cpp;; *) echo "bad tables" >&2; exit 2;; esac
mkdir -p "$OUT"
TMP="$(mktemp -d /tmp/ltref.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
for f in "$REF"/*.cpp "$REF"/*.h; do
  b="$(basename "$f")"
  [ "$b" = "LightTrack.h" ] && continue
  ln -s "$f" "$TMP/$b"
done
cp "$HERE/shadow/LightTrack.h" "$TMP/LightTrack.h"
SRCS=""
for s in Clipp2D Clipp3D Colour EngBase EngPoly EngBcub GraphicsBase GraphicsPoly TransLn data; do SRCS="$SRCS $TMP/$s.cpp"; done
SRCS="$SRCS $TMP/$TFILE"
g++ -shared -fPIC -O2 -w -fpermissive -fwrapv \
    -D_RGB_ -D_32BPP_ -D_FIXED_ -D_Z_BUFFER_ -D_MONO_ \
    -DHW_SCREEN_X_SIZE="$W" -DHW_SCREEN_Y_SIZE="$H" \
    -I"$TMP" $SRCS "$HERE/ref_driver.cpp" -o "$OUT/libltref_${W}x${H}_${TAB}.so"
echo "built $OUT/libltref_${W}x${H}_${TAB}.so"
