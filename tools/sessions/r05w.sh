#!/bin/bash
# config E layer: round-4 tree (.abtree/old) against the current one, same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  for v in old new; do
    D=$GRAFT_REPO_ROOT; [ $v = old ] && D=$GRAFT_REPO_ROOT/.abtree/old
    ( cd $D && timeout 200 python tools/run_e.py --nograph 2>/dev/null | tail -1 | sed "s/^/$v $rep /" )
  done
done
