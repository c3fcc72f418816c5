import os, random, subprocess, sys, gzip
G='/root/repo/tests/golden'
seeds = {"gtf":[G+"/cse_ref/test_ensemble_chr22.2.gtf"], "vcf":[G+"/cse_ref/test1.vcf"], "bed":[G+"/annot_ref/junctions_extract.bed"], "fa":[G+"/cse_ref/test_chr22.fa"]}
rng=random.Random(3)
bad=0; runs=0
for kind, files in seeds.items():
    data=open(files[0],'rb').read()
    if kind=="fa": data=data[:200000]
    if kind=="gtf": data=data[:400000]
    for it in range(int(sys.argv[1]) if len(sys.argv)>1 else 150):
        b=bytearray(data)
        m=rng.random()
        if m<0.35:
            for _ in range(rng.choice([1,3,20])): b[rng.randrange(len(b))]=rng.choice([9,10,32,48,45,59,34,0,255,rng.randrange(256)])
        elif m<0.55: b=b[:rng.randrange(len(b)+1)]
        elif m<0.7:
            k=rng.randrange(len(b)); b[k:k]=bytes(rng.choice([9,10,59,34,32]) for _ in range(rng.randrange(1,40)))
        elif m<0.85:
            lines=b.split(b"\n"); rng.shuffle(lines); b=bytearray(b"\n".join(lines[:rng.randrange(1,len(lines)+1)]))
        else:
            lines=b.split(b"\n"); k=rng.randrange(len(lines)); f=lines[k].split(b"\t")
            if f: f[rng.randrange(len(f))]=rng.choice([b"",b"-1",b"99999999999999999999",b"1e9",b"chr",b"\xff\xfe"]); lines[k]=b"\t".join(f)
            b=bytearray(b"\n".join(lines))
        p="/tmp/csefuzz/in."+kind
        if kind=="vcf" and rng.random()<0.3:
            p+=".gz"; open(p,'wb').write(gzip.compress(bytes(b)) if rng.random()<0.7 else gzip.compress(bytes(b))[:rng.randrange(1,200)])
        else: open(p,'wb').write(bytes(b))
        if kind=="fa" and os.path.exists(p+".fai"): os.remove(p+".fai")
        r=subprocess.run(["/tmp/csefuzz/harness",kind,p],capture_output=True,timeout=60)
        runs+=1
        if r.returncode not in (0,) or b"runtime error" in r.stderr or b"AddressSanitizer" in r.stderr:
            bad+=1; print(kind,it,r.returncode,r.stderr[-600:].decode(errors="replace")); 
            open("/tmp/csefuzz/crash_%s_%d"%(kind,it),'wb').write(bytes(b))
            if bad>5: sys.exit(1)
print("runs",runs,"bad",bad)
