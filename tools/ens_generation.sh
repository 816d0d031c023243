# tools/ens_generation.py under the three knob settings (round-1 serial loop / step graphs concurrent / persistent kernel)
cd $GRAFT_REPO_ROOT
for preset in ddl_ray3 ddl_normal3; do
for knobs in "DDL_MLP_PERSISTENT=0 DDL_SERIAL_MEMBERS=1 DDL_CHUNK=1" "DDL_MLP_PERSISTENT=0" "DDL_MLP_PERSISTENT=1"; do
  env $knobs timeout -k 10 280 python -u tools/ens_generation.py --preset $preset 2>&1 | tail -1 || exit 1
done; done
