#!/bin/bash
# Everything the round-end driver does, in one go, on an MI355X box:  bash tools/run_all_checks.sh
# (build -> CPU suite -> GPU parity suite -> smoke -> bench line -> inference bench).  Stops at the first failure.
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py
python -m pytest tests -x -q -m "not gpu"
python -m pytest tests -x -q -m gpu
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 10 --warmup 3
python bench.py --config rice416-bf16 --steps 20
