import json, sys, numpy as np
sys.path.insert(0,'/root/repo')
import ozimmu_amd as oz
rows=[json.loads(l) for f in ["gpurun_out/r6j/policy_q3_s9_seed11.jsonl","gpurun_out/r6j/policy_q3_s9_shortk_seed12.jsonl"] for l in open(f)]
rows=[r for r in rows if r["ran"].get("k64_breg")=="k64_breg" and r["ran"].get("k64")=="k64"]
print(len(rows),"rows with both k64 forms")
P0=oz.policy_params()
names=["K2_A","K2_BETA","K2_F","K2_STEP","CL_A","CL_A1","CL_BETA","CL_B","CL_F","CL_STEP","CL4_A","CL4_STEP","W_A","W_BETA","W_B","W_STEP","X_A","X_BETA","X_B","X_STEP","Y_A","Y_BETA","Y_B","Y_STEP","Z_A","Z_BETA","Z_B","Z_STEP"]
iz=names.index("Z_A")
def preds(params):
    oz.policy_params(params)
    out=[]
    for r in rows:
        # device of the data: set via policy_predict_device? use handle-less nominal; scale by mfma ratio below
        p,_=oz.policy_predict(None,9,r["m"],r["n"],r["k"])
        out.append(p)
    return out
base=preds(P0)
def ratio(kern,pp): return np.array([r["us"][kern]/p[kern] for r,p in zip(rows,pp) if kern in p])
for kern in ("classic","wide","k64","k64_breg"):
    x=ratio(kern,base); print(kern,"meas/pred median %.3f  p10 %.3f p90 %.3f"%(np.median(x),np.percentile(x,10),np.percentile(x,90)))
f=np.median(ratio("k64",base))   # box factor from the unchanged sibling kernel
from scipy.optimize import minimize
def loss(z):
    P=list(P0); P[iz]=z[0]; P[iz+2]=z[1]; P[iz+3]=z[2]
    pp=preds(P)
    x=np.log(ratio("k64_breg",pp)/f)
    return float(np.mean(x*x))
z0=[P0[iz],P0[iz+2],P0[iz+3]]
print("start",z0,loss(z0))
res=minimize(loss,z0,method="Nelder-Mead",options={"xatol":1e-3,"fatol":1e-6,"maxiter":300})
print("fit Z_A,Z_B,Z_STEP",res.x,res.fun)
P=list(P0); P[iz]=res.x[0]; P[iz+2]=res.x[1]; P[iz+3]=res.x[2]
pp=preds(P)
# regret old vs new on these rows (box-scaled times are measured ones)
def regret(pp):
    tot=[]
    for r,p in zip(rows,pp):
        cand={k:v for k,v in p.items() if k in r["us"]}
        pick=min(cand,key=cand.get)
        pk="k64_breg" if pick=="k64_breg" else pick
        best=min(r["us"].values())
        tot.append(r["us"][pk]/best-1)
    return np.mean(tot)*100,np.percentile(tot,90)*100,max(tot)*100
print("regret old mean/p90/max %",regret(base)); print("regret new",regret(pp))
oz.policy_params(P0)
print("---- Z_B sweep (Z_A, Z_STEP as compiled)")
for zb in (5.304,4.5,4.0,3.5,3.0,2.5,2.0):
    for za in (0.7466,0.76,0.78):
        P=list(P0); P[iz+2]=zb; P[iz]=za
        pp=preds(P)
        x=np.log(ratio("k64_breg",pp)/f)
        print("Z_A %.4f Z_B %.2f  rms log err %.4f  regret mean %.2f p90 %.2f max %.1f"%((za,zb,float(np.sqrt(np.mean(x*x))))+regret(pp)))
oz.policy_params(P0)
