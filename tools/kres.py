"""Registers, scratch, occupancy and LDS of every kernel in one source, with the product build's flags:
    python tools/kres.py mjmpc_amd/csrc/tree_rollout.hip [name filter] [extra flags]"""
import os,re,sys,subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjmpc_amd.build import flags_for
for alt in range(3):        # (the product build's alternatives: the first flag set this compiler takes)
    res=subprocess.run(["/opt/rocm/bin/hipcc"]+flags_for(sys.argv[1],alt)+["-Rpass-analysis=kernel-resource-usage","-c",sys.argv[1],"-o","/tmp/x.o"]+sys.argv[3:],capture_output=True,text=True)
    if res.returncode==0: break
out=res.stderr
cur=None; rows={}
for ln in out.splitlines():
    m=re.search(r"Function Name: (\S+)",ln)
    if m:
        d=subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()
        d=d.replace("mjmpc::(anonymous namespace)::","").replace("void ","")
        cur=d.split("(")[0]; rows[cur]={}
    for k in ("VGPRs","AGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","SGPRs Spill","VGPRs Spill","LDS Size [bytes/block]"):
        m=re.search(r"remark:\s+"+re.escape(k)+r": (\d+)",ln)
        if m and cur: rows[cur][k]=int(m.group(1))
    if " error" in ln: print(ln)
filt=sys.argv[2] if len(sys.argv)>2 else ""
for k,v in rows.items():
    if filt in k: print("%-60s"%k, " ".join("%s=%d"%(a.split(" [")[0].replace(" ",""),b) for a,b in v.items()))
