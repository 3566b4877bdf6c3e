rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -1
python3 -c "from clonealign_amd import engine as E; print('ca_build_id', E.build_id())"
git_rev=$(cat .git_rev 2>/dev/null); echo "sources: $git_rev"
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "== __graft_entry__.smoke()"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
