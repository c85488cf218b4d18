import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from common import zoo, voice_signal
from obs_rvc_amd.rvc import RvcInfer
from obs_rvc_amd.resample import FftFixedInOut
from obs_rvc_amd.streaming import NativeStreamingSession
from oracle import resample_oracle as RO
z=zoo("tiny"); e=RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(); e.load_model(z["model"])
for ri,ro,ch in [(44100,48000,10000),(48000,47999,1000),(16000,48000,1),(48000,16000,3),(8000,96000,500),(96000,8000,6000),(22050,16000,2205),(48000,48000,19200),(48000,48000,19201),(11025,44100,4000)]:
    try:
        r=FftFixedInOut(e,ri,ro,ch); fi,fo=r.input_frames_next(),r.output_frames_max()
        o=RO.FftFixedInOut(ri,ro,ch)
        x=voice_signal(fi*2,seed=1)
        err=max(float(np.abs(r.process(x[i*fi:(i+1)*fi])-o.process(x[i*fi:(i+1)*fi])).max()) for i in range(2))
        print("resampler",ri,ro,ch,"->",fi,fo,"err %.2e"%err)
    except Exception as ex:
        print("resampler",ri,ro,ch,"RAISED",str(ex)[:70])
for args in [(48000,0.01,0.07,2.0,4800),(48000,0.16,0.0,2.0,4800),(48000,0.16,0.07,0.0,4800),(48000,1.0,0.07,2.0,4800),(16000,0.16,0.07,2.0,4800),(48000,0.16,0.5,0.1,4800),(32000,0.2,0.05,1.0,4800)]:
    try:
        s=NativeStreamingSession(e,*args,12,0.5); F=s.sample_frame_size
        ys=[s.process_one_frame(voice_signal(F,seed=c)) for c in range(4)]
        print("session",args,"frame",F,"R",s.model_return_length,"finite",bool(np.isfinite(ys[-1]).all()))
    except Exception as ex:
        print("session",args,"RAISED",str(ex)[:80])
print("after", bool(np.isfinite(e.infer(voice_signal(35840,seed=1),2560,12,200,21)).all()))
