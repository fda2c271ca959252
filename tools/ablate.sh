cp kbo_amd/libkbo_hip.so /tmp/good.so
for A in 0 1 2 4 7; do
  if [ $A != 0 ]; then cp kbo_amd/libkbo_hip_ab$A.so kbo_amd/libkbo_hip.so; else cp /tmp/good.so kbo_amd/libkbo_hip.so; fi
  echo "== ablate=$A (1=no stores 2=no second rank load 4=no query refetch)"
  ONLY=1 G=${G:-5000000} timeout 200 python tools/sweep_walk.py 2>&1 | grep only
done
