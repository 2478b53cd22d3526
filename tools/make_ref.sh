#!/bin/sh
# Materialise reference emcee where bench.py's cpu_baseline leg can time IT (not a port) on the GPU box's host cores.
#
# /root/reference does not travel to the GPU box, but git-ignored files of this tree do (libemx.so does): this recipe copies the
# reference's Python package -- verbatim, unmodified -- into the git-ignored oracle/_ref/emcee, plus the `emcee_version` module
# its setup.py generates at build time (setup.py:59-64; absent from the source tree).  Nothing under oracle/_ref is ever
# committed, and nothing under emcee_amd/ may import it (tests/test_layout.py); only tests/, smoke() and bench.py's
# cpu_baseline leg reach it, through oracle/ref_shim.py.  Run by __graft_entry__.build() whenever /root/reference is present.
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
SRC="${1:-/root/reference/src/emcee}"
DST="$HERE/oracle/_ref"
[ -d "$SRC" ] || { echo "make_ref: $SRC not present (nothing to do)"; exit 0; }
rm -rf "$DST/emcee"
mkdir -p "$DST/emcee"
# sources only: no bytecode, no tests (the tests import pytest fixtures this leg never needs)
(cd "$SRC" && find . -name '*.py' -not -path './tests/*' -not -path '*/__pycache__/*' | while read -r f; do
    mkdir -p "$DST/emcee/$(dirname "$f")"
    cp "$f" "$DST/emcee/$f"
done)
printf '__version__ = "0+reference"\n' > "$DST/emcee/emcee_version.py"
printf 'materialised from %s by tools/make_ref.sh; git-ignored, test/benchmark infrastructure only\n' "$SRC" > "$DST/README"
echo "make_ref: $(find "$DST/emcee" -name '*.py' | wc -l) files -> $DST/emcee"
