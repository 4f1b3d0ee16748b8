#!/bin/bash
# The reference's OWN tests and example renderers, unmodified, against the `nerfacc` alias of this
# repository.  Nothing of the reference is committed: `prepare` copies the files
# from /root/reference into build/ref_suite/ (git-ignored; it travels to the GPU box with gpurun),
# `run` executes them there with PYTHONPATH pointing at this repository.
#
#   tools/run_reference_suite.sh prepare          # build container (has /root/reference)
#   tools/run_reference_suite.sh run [out.md]     # GPU box (no /root/reference needed)
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
DST="$ROOT/build/ref_suite"
REF="${NERFACC_REFERENCE:-/root/reference}"

case "${1:-run}" in
prepare)
    [ -d "$REF/tests" ] || { echo "no reference tree at $REF" >&2; exit 1; }
    rm -rf "$DST"; mkdir -p "$DST/tests" "$DST/examples/datasets"
    for t in rendering scan grid pack pdf; do cp "$REF/tests/test_$t.py" "$DST/tests/"; done
    cp "$REF/examples/utils.py" "$DST/examples/utils.py"
    cp "$REF/examples/datasets/__init__.py" "$REF/examples/datasets/utils.py" "$DST/examples/datasets/"
    (cd "$DST" && sha256sum tests/*.py examples/utils.py examples/datasets/*.py) > "$DST/SHA256SUMS"
    (cd "$REF" && for f in tests/test_{rendering,scan,grid,pack,pdf}.py examples/utils.py examples/datasets/__init__.py examples/datasets/utils.py; do sha256sum "$f"; done) > "$DST/SHA256SUMS.reference"
    diff <(awk '{print $1}' "$DST/SHA256SUMS" | sort) <(awk '{print $1}' "$DST/SHA256SUMS.reference" | sort) \
        && echo "prepared $DST (byte-identical copies of the reference's files)"
    ;;
run)
    OUT="$(realpath -m "${2:-$ROOT/gpurun_out/r03_reference_suite.md}")"
    mkdir -p "$(dirname "$OUT")"
    RC="$(mktemp)"; echo 0 > "$RC"
    [ -d "$DST/tests" ] || { echo "run '$0 prepare' where /root/reference exists first" >&2; exit 1; }
    cd "$DST"
    {
        echo "# The reference's own tests and example renderers on MI355X against the \`nerfacc\` alias"
        echo
        echo "Files: byte-identical copies of \`/root/reference/tests/test_{rendering,scan,grid,pack,pdf}.py\`,"
        echo "\`examples/utils.py\`, \`examples/datasets/{__init__,utils}.py\` (sha256 below), run with"
        echo "\`PYTHONPATH=<this repo>\` so that \`import nerfacc\` resolves to \`nerfacc/\` -> \`nerfacc_amd\`."
        echo
        echo '```'
        cat SHA256SUMS
        echo '```'
        echo
        echo "## pytest (the reference's tests, unmodified)"
        echo
        echo '```'
        PYTHONPATH="$ROOT" python -m pytest tests -v -p no:cacheprovider --rootdir "$DST" -c /dev/null > "$RC.log" 2>&1 || echo 1 > "$RC"
        grep -v "^$" "$RC.log" | tail -n 60
        echo '```'
        echo
        echo "## The reference's examples/utils.py renderers on the bench scene"
        echo
        PYTHONPATH="$ROOT" python -W ignore "$ROOT/tools/ref_examples_check.py" 2> "$RC.err" || { echo 1 > "$RC"; tail -n 20 "$RC.err"; }
    } | tee "$OUT"
    rc="$(cat "$RC")"; rm -f "$RC" "$RC.log" "$RC.err"
    exit "$rc"
    ;;
*) echo "usage: $0 prepare|run [out.md]" >&2; exit 2 ;;
esac
