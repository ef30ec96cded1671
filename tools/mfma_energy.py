import subprocess, sys, time, threading, re, os, glob
# sample board power / sclk via hwmon of the GPU in use while each variant runs
def find_hwmon():
    best=None
    for h in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
        if os.path.exists(h+'/power1_average') or os.path.exists(h+'/power1_input'): best=h if best is None else best
    return best
def read(p):
    try: return int(open(p).read())
    except Exception: return None
for which,name in [(1,'16x16x64 1w'),(2,'32x32x32 1w'),(3,'16x16x64 2w'),(4,'32x32x32 2w')]:
    samples=[]
    stop=False
    def samp():
        hs=glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')
        while not stop:
            best=(0,None)
            for h in hs:
                p=read(h+'/power1_average') or read(h+'/power1_input') or 0
                f=read(h+'/freq1_input') or 0
                if p>best[0]: best=(p,f)
            samples.append(best); time.sleep(0.1)
    t=threading.Thread(target=samp); t.start()
    out=subprocess.run(['./tools/ubench_mfma_energy','3',str(which)],capture_output=True,text=True).stdout.strip()
    stop=True; t.join()
    late=samples[len(samples)//2:]
    pw=sum(s[0] for s in late)/max(1,len(late))/1e6; fq=sum((s[1] or 0) for s in late)/max(1,len(late))/1e6
    print(out, f"| board power {pw:.0f} W  sclk {fq:.0f} MHz (busiest card, {len(late)} samples)", flush=True)
