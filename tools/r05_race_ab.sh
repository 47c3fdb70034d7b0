#!/bin/bash
# A/B that pins the halo-header race of round 5: the full GPU suite (the 8-rank tests fail only inside it) with one step-message header per inbox
# side (SPH_TEST_SINGLE_HEADER=1: the protocol until round 5) and with one per message parity (default)
cd ${GRAFT_REPO_ROOT:-.}
for v in "SPH_TEST_SINGLE_HEADER=1" "SPH_TEST_SINGLE_HEADER=1" "X=1" "X=1"; do
  echo "== $v"; env $v python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^rank [0-9]:|FAILED" | cut -c1-200
done
