import json,sys
d=json.load(open(sys.argv[1]))
print({k:d[k] for k in ('value','ms_per_step','steps')})
r=d['roofline']
print("roof", {k:r.get(k) for k in ('timing_class','achieved','frac','avg_launch_ms','whole_path_tflops','whole_path_frac','traffic')})
print("kernels", {k:round(v,1) for k,v in d['kernels_ms_per_step'].items()})
ff=d.get('fit_forecast')
if ff:
    for name,leg in ff['legs'].items():
        ce=leg.get('cpu_estimate',{})
        print(f"  {name:32s} gpu_s {leg['gpu_s']:8.2f}  cpu_est_s {ce.get('wall_s_on_usable_cores',0):9.1f} ({ce.get('cores_used')} cores)  x{leg.get('speedup_vs_cpu_estimate',0):.1f}")
    print("  again_s", ff['forecast_with_nowcasts_again_s'])
oc=d.get('other_configs')
if oc:
    for k,v in oc.items():
        print(" ", k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('value','ms_per_step','roofline_frac','roofline_kernel_class','fp64_path_ms_per_step','max_rel_logml_diff_vs_fp64_path','wall_s_with_startup','error')})
cb=d.get('cpu_baseline')
if cb: print("cpu_baseline", cb.get('value'), cb.get('cores'), d.get('speedup_vs_cpu_port'))
