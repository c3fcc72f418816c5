"""What the annotated-VCF path SAYS (tests/hostemu's emu_vcf_rewrite in a subprocess per input) against what the real reference says (the htslib part of
`regtools_ref variants annotate -o`'s stderr, __FILE__ cut down to the file's name), on the inputs of tests/vcf_cases.py with a few bytes flipped / deleted /
doubled / inserted; also whether both end with a status of zero.  Dev container only.
    python tools/fuzz/vcf_stderr_diff.py N SEED"""
import subprocess,sys,re,os,random,tempfile
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,os.path.join(ROOT,'tests'))
import vcf_cases
n,seed=int(sys.argv[1]),int(sys.argv[2])
RUNNER="""
import ctypes,sys
lib=ctypes.CDLL(sys.argv[1])
err=ctypes.create_string_buffer(512)
rc=lib.emu_vcf_rewrite(sys.argv[2].encode(),sys.argv[3].encode(),err,512)
sys.stderr.flush()
print("RC",rc,err.value)
"""
rng=random.Random(seed)
def mutate(data):
    b=bytearray(data)
    for _ in range(rng.choice([1,1,2,4])):
        if not b: break
        k,p=rng.random(),rng.randrange(len(b))
        pool=[9,10,ord(":"),ord(";"),ord(","),ord("="),ord("."),ord("<"),ord(">"),ord('"'),ord("|"),ord("/"),ord(" "),ord("-"),ord("1"),ord("e"),ord("\\")]
        if k<0.5: b[p]=rng.choice(pool)
        elif k<0.65: del b[p:p+rng.randrange(1,30)]
        elif k<0.8: b[p:p]=b[p:p+rng.randrange(1,30)]
        else: b[p:p]=bytes(rng.choice([b"\t",b"\t\t",b":",b";;",b"=",b",,",b"\n",b"\t.\t",b":.",b"./."]))
    return bytes(b)
bad=0
with tempfile.TemporaryDirectory() as td:
    inputs=vcf_cases.build(td)
    texts=sorted(k for k in inputs if not k.endswith("_gz") and "bcf" not in k and not k.startswith("tbi"))
    src=os.path.join(td,'in.vcf'); open('/tmp/far.gtf','w').write(vcf_cases.GTF_FAR)
    for k in range(n):
        open(src,'wb').write(mutate(inputs[rng.choice(texts)]))
        a=subprocess.run([sys.executable,'-c',RUNNER,os.path.join(ROOT,'tests','hostemu','libhostemu.so'),src,td+'/n.vcf'],capture_output=True)
        try: b=subprocess.run([os.path.join(ROOT,'oracle','_ref','regtools_ref'),'variants','annotate','-o',td+'/r.vcf',src,'/tmp/far.gtf'],capture_output=True,timeout=20)
        except subprocess.TimeoutExpired: continue
        if b.returncode not in (0,1): continue
        ea=[l for l in a.stderr.decode('latin1').split('\n') if l]
        eb=b.stderr.decode('latin1').split('\n')
        eb=[re.sub(r'\[/root/reference/src/utils/htslib/(vcf\.c:\d+ )',r'[\1',l) for l in eb]
        i=max(j for j,l in enumerate(eb) if l.startswith('Output file'))+2
        eb=[l for l in eb[i:] if l]
        m=re.search(r"RC (\d+) b'(.*)'",a.stdout.decode('latin1'))
        rca,msg=int(m.group(1)),m.group(2)
        if rca in (2,3,4,5) and msg: ea.append(msg.replace('\\n',''))
        if rca==1 and msg: ea+= [l for l in msg.encode().decode('unicode_escape').split('\n') if l]
        if ea!=eb or (rca!=0)!=(b.returncode!=0):
            bad+=1
            if bad<=12:
                keep='/tmp/se_%d_%d.vcf'%(seed,k); open(keep,'wb').write(open(src,'rb').read())
                print('DIFF',k,rca,b.returncode,keep)
                import difflib
                for l in list(difflib.unified_diff(ea,eb,lineterm='',n=0))[:8]: print('   ',l[:200])
print('runs',n,'stderr differences',bad)
