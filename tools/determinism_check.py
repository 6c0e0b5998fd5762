"""Repeats the whole forward 40 times per shape and requires bitwise identical results (catches races and hazards that a tolerance hides)."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import applied_image_processing_amd.synth as synth
from applied_image_processing_amd.AdaIN import net, test as t
vgg_sd = synth.to_torch(synth.vgg_state_dict(0, full=True)); dec_sd = synth.to_torch(synth.decoder_state_dict(0))
net.vgg.load_state_dict(vgg_sd); net.decoder.load_state_dict(dec_sd); net.vgg.to("cuda:0"); net.decoder.to("cuda:0")
bad = 0
for (h, w, n) in ((1024, 1024, 1), (1080, 1920, 2), (517, 333, 3), (2048, 2048, 1), (1200, 1600, 2), (256, 456, 4)):     # the last two run the 16 x 16 tile geometry
    c = torch.from_numpy(synth.image(1, n, h, w)).cuda(); s = torch.from_numpy(synth.image(2, n, 512, 512)).cuda()
    ref = t.style_transfer_simple(net.vgg, net.decoder, c, s, 0.5)
    for i in range(40):
        out = t.style_transfer_simple(net.vgg, net.decoder, c, s, 0.5)
        if not torch.equal(out, ref):
            bad += 1; print("MISMATCH", h, w, n, i, float((out - ref).abs().max()))
    print(h, w, n, "40 repeats identical" if not bad else "differences", flush=True)
# round 6: the latency schedule (cin split + fixed-order combine) on single small frames - the same call must give the same bits
import applied_image_processing_amd.runtime as rt
with rt.schedule(rt.SCHEDULE_LATENCY):
    for (h, w, n) in ((256, 456, 1), (128, 228, 1), (270, 480, 1), (64, 96, 2), (512, 512, 1)):
        c = torch.from_numpy(synth.image(3, n, h, w)).cuda(); s = torch.from_numpy(synth.image(2, n, 512, 512)).cuda()
        ref = t.style_transfer_simple(net.vgg, net.decoder, c, s, 0.5)
        for i in range(40):
            out = t.style_transfer_simple(net.vgg, net.decoder, c, s, 0.5)
            if not torch.equal(out, ref):
                bad += 1; print("MISMATCH (latency schedule)", h, w, n, i, float((out - ref).abs().max()))
        print(h, w, n, "latency schedule: 40 repeats identical" if not bad else "differences", flush=True)
print("determinism:", "ok" if not bad else f"{bad} mismatches")
sys.exit(1 if bad else 0)
