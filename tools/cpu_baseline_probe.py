import sys, time, json
sys.path.insert(0, '.')
import bench
for n in (50, 130, 171):
    t=time.time(); r=bench.cpu_baseline(n_per_dim=n, steps=10, budget_s=300); print(n, round(time.time()-t,1), 's', json.dumps(r))
