"""Generates the classifier golden fixtures with the CPU oracle in exact mode
(oracle/fav_exact.c order; bit-reproducible).  Inputs are regenerated from seeds, so
the fixtures hold only expected outputs.

  python tests/golden/make_classifier_fixtures.py mc       # 64 frames, T=30 all_blocks            (~3 min on 8 AVX-512 cores)
  python tests/golden/make_classifier_fixtures.py 10k      # 10,000 frames, single pass            (~10 min)
  python tests/golden/make_classifier_fixtures.py mfma_mc  # 64 frames, T=30 all_blocks, production bf16-MFMA model (~15 min)
  python tests/golden/make_classifier_fixtures.py mfma_mc512 # 512 frames (two headline batches), T=30 all_blocks, production model (~1.5-2.5 h; resumes)
  python tests/golden/make_classifier_fixtures.py mfma_10k # 10,000 frames, single pass, production bf16-MFMA model (~1 h; resumes)
  python tests/golden/make_classifier_fixtures.py vit     # ViT-B/16, 64 corrupted frames, production bf16-MFMA model
  python tests/golden/make_classifier_fixtures.py ens5    # BASELINE configs[3]: 5 ResNet-50 members (seeds 1..5), 256 frames, production model (~8 min)
"""
import os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from failure_aware_vision_amd import synth, weights
from oracle import fav_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FRAME_SEED, NOISE_SEED, SEVERITY = 21, 3, 3
what = sys.argv[1] if len(sys.argv) > 1 else "mc"
if what == "vit":
    blob, info = weights.make_synthetic_vit("vit_b16", seed=1)
else:
    blob, info = weights.make_synthetic("resnet50", seed=1)
model = O.parse_blob(blob)


def frames(start, n):
    u8 = synth.synthetic_frames_u8(n, 224, 224, seed=FRAME_SEED, start_id=start)
    return synth.gaussian_noise_f32(u8, SEVERITY, seed=NOISE_SEED, start_id=start)


def frame_crc(lg):
    """zlib.crc32 of each frame's fp32 logits [T, classes] (little endian, t-major)."""
    lg = np.asarray(lg, np.float32)
    return np.array([zlib.crc32(np.ascontiguousarray(lg[:, i, :]).tobytes()) for i in range(lg.shape[1])], np.uint32)


def gap_of(pbar):
    s = np.sort(pbar, axis=1)
    return (s[:, -1] - s[:, -2]).astype(np.float32)


if what == "vit":
    n = 64                                  # the config's per-GPU share
    cfg = O.ClassifyConfig(exact="mfma", temperature=1.5, conf_kind=O.CONF_ENTROPY)
    labels, conf, gaps, crcs = [], [], [], []
    for s in range(0, n, 4):
        t0 = time.time()
        l, c, lg, pb = O.classify(model, frames(s, 4), cfg, return_logits=True)
        labels.append(l); conf.append(c); gaps.append(gap_of(pb)); crcs.append(frame_crc(lg))
        print("vit", s, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(HERE, "vit_b16_mfma_64.npz"), labels=np.concatenate(labels).astype(np.int16),
                        conf=np.concatenate(conf), gap=np.concatenate(gaps), logit_crc32=np.concatenate(crcs),
                        blob_sha256=info["sha256"],
                        meta="vit_b16 seed1; frames seed 21 ids 0..63 + gaussian noise sev3 seed 3; single pass, entropy "
                             "confidence at temperature 1.5; production mode (v_mfma_f32_16x16x32_bf16 model)")
elif what == "ens5":
    # BASELINE configs[3]: five independently seeded ResNet-50 members, 224x224, severity-3 frames, single pass per
    # member, head = mean over members of softmax.  Per member and frame: CRC-32 of the 1000 fp32 logits.
    n, seeds = 256, (1, 2, 3, 4, 5)        # the config's global batch
    cfg = O.ClassifyConfig(exact="mfma")
    x = frames(0, n)
    lgs, shas = [], []
    for sd in seeds:
        t0 = time.time()
        mb, mi = weights.make_synthetic("resnet50", seed=sd)
        shas.append(mi["sha256"])
        mm = O.parse_blob(mb)
        parts = [O.classify(mm, x[s:s + 4], cfg, return_logits=True)[2] for s in range(0, n, 4)]
        lgs.append(np.concatenate(parts, axis=1)[0])          # [n, 1000]
        print("ens5 member seed", sd, time.time() - t0, flush=True)
    lg = np.stack(lgs)                                          # [5, n, 1000]: the head sees members as samples
    l, c, pb = O.confidence_head(lg)
    crc = np.array([[zlib.crc32(np.ascontiguousarray(lg[m, i]).tobytes()) for i in range(n)] for m in range(len(seeds))], np.uint32)
    np.savez_compressed(os.path.join(HERE, "r50_ens5_mfma_256.npz"), labels=l.astype(np.int16), conf=c, gap=gap_of(pb),
                        member_logit_crc32=crc, member_blob_sha256=np.array(shas), member_seeds=np.array(seeds),
                        member_labels=lg.argmax(axis=2).astype(np.int16), blob_sha256=shas[0],
                        meta="5 x resnet50 seeds 1..5; frames seed 21 ids 0..255 + gaussian noise sev3 seed 3; single pass per member, "
                             "mean of member softmax; production mode (v_mfma_f32_16x16x32_bf16 model); member_logit_crc32[m][i] = "
                             "zlib.crc32 of member m's 1000 fp32 logits of frame i")
elif what == "mc":
    n, T = 64, 30
    cfg = O.ClassifyConfig(n_samples=T, site_mask=weights.site_mask_for(1, "all_blocks"), p=0.1, seed=4, exact=True)
    labels, conf, gaps = [], [], []
    for s in range(0, n, 8):
        t0 = time.time()
        l, c, lg, pb = O.classify(model, frames(s, 8), cfg, img_ids=np.arange(s, s + 8), return_logits=True)
        labels.append(l); conf.append(c); gaps.append(gap_of(pb))
        print("mc", s, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(HERE, "r50_exact_mc30_64.npz"), labels=np.concatenate(labels).astype(np.int16),
                        conf=np.concatenate(conf), gap=np.concatenate(gaps), blob_sha256=info["sha256"],
                        meta="resnet50 seed1; frames seed 21 ids 0..63 + gaussian noise sev3 seed 3; T=30 all_blocks p=0.1 seed 4; exact")
elif what == "mfma_mc":
    n, T = 64, 30
    cfg = O.ClassifyConfig(n_samples=T, site_mask=weights.site_mask_for(1, "all_blocks"), p=0.1, seed=4, exact="mfma")
    labels, conf, gaps, logits = [], [], [], []
    for s in range(0, n, 4):
        t0 = time.time()
        l, c, lg, pb = O.classify(model, frames(s, 4), cfg, img_ids=np.arange(s, s + 4), return_logits=True)
        labels.append(l); conf.append(c); gaps.append(gap_of(pb)); logits.append(frame_crc(lg))
        print("mfma_mc", s, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(HERE, "r50_mfma_mc30_64.npz"), labels=np.concatenate(labels).astype(np.int16),
                        conf=np.concatenate(conf), gap=np.concatenate(gaps), logit_crc32=np.concatenate(logits),
                        blob_sha256=info["sha256"],
                        meta="resnet50 seed1; frames seed 21 ids 0..63 + gaussian noise sev3 seed 3; T=30 all_blocks p=0.1 seed 4; "
                             "production mode (v_mfma_f32_16x16x32_bf16 model)")
elif what == "mfma_mc512":
    # the HEADLINE config on two of its own batches (512 frames, T = 30, all_blocks): per-frame CRC-32 of all 30 x 1000 logits
    n, T, bs = 512, 30, 4
    part = os.path.join(HERE, "_mfma_mc512_partial.npz")
    done = 0
    labels = np.zeros(n, np.int16); conf = np.zeros(n, np.float32); gaps = np.zeros(n, np.float32); lsum = np.zeros(n, np.uint32)
    if os.path.exists(part):
        d = np.load(part)
        if str(d["blob_sha256"]) == info["sha256"]:
            done = int(d["done"]); labels, conf, gaps, lsum = d["labels"], d["conf"], d["gap"], d["lsum"]
    cfg = O.ClassifyConfig(n_samples=T, site_mask=weights.site_mask_for(1, "all_blocks"), p=0.1, seed=4, exact="mfma")
    t0 = time.time()
    for s in range(done, n, bs):
        l, c, lg, pb = O.classify(model, frames(s, bs), cfg, img_ids=np.arange(s, s + bs), return_logits=True)
        labels[s:s + bs] = l; conf[s:s + bs] = c; gaps[s:s + bs] = gap_of(pb); lsum[s:s + bs] = frame_crc(lg)
        np.savez(part, done=s + bs, labels=labels, conf=conf, gap=gaps, lsum=lsum, blob_sha256=info["sha256"])
        print("mfma_mc512", s + bs, round(time.time() - t0, 1), flush=True)
    np.savez_compressed(os.path.join(HERE, "r50_mfma_mc30_512.npz"), labels=labels, conf=conf, gap=gaps, logit_crc32=lsum,
                        blob_sha256=info["sha256"],
                        meta="resnet50 seed1; frames seed 21 ids 0..511 + gaussian noise sev3 seed 3; T=30 all_blocks p=0.1 seed 4; "
                             "production mode (v_mfma_f32_16x16x32_bf16 model); logit_crc32 = zlib.crc32 of each frame's [30][1000] fp32 logits")
    if os.path.exists(part):
        os.remove(part)
elif what == "mfma_10k":
    # north_star's "label-exact agreement on 10k corrupted test frames" in the arithmetic that ships: the same 10,000
    # frames as the exact-mode fixture through the bit-exact model of v_mfma_f32_16x16x32_bf16
    n, bs = 10000, 50
    part = os.path.join(HERE, "_mfma10k_partial.npz")
    done = 0
    labels = np.zeros(n, np.int16); conf = np.zeros(n, np.float32); gaps = np.zeros(n, np.float32)
    lsum = np.zeros(n, np.uint32)
    if os.path.exists(part):
        d = np.load(part)
        if str(d["blob_sha256"]) == info["sha256"]:
            done = int(d["done"]); labels, conf, gaps, lsum = d["labels"], d["conf"], d["gap"], d["lsum"]
    cfg = O.ClassifyConfig(exact="mfma")
    t0 = time.time()
    for s in range(done, n, bs):
        l, c, lg, pb = O.classify(model, frames(s, bs), cfg, return_logits=True)
        labels[s:s + bs] = l; conf[s:s + bs] = c; gaps[s:s + bs] = gap_of(pb)
        lsum[s:s + bs] = frame_crc(lg)
        if (s // bs) % 4 == 3:
            np.savez(part, done=s + bs, labels=labels, conf=conf, gap=gaps, lsum=lsum, blob_sha256=info["sha256"])
            print("mfma_10k", s + bs, round(time.time() - t0, 1), flush=True)
    np.savez_compressed(os.path.join(HERE, "r50_mfma_10k_noise3.npz"), labels=labels, conf=conf, gap=gaps, logit_crc32=lsum,
                        blob_sha256=info["sha256"],
                        meta="resnet50 seed1; frames seed 21 ids 0..9999 + gaussian noise sev3 seed 3; single pass; "
                             "production mode (v_mfma_f32_16x16x32_bf16 model); logit_crc32 = zlib.crc32 of each frame's fp32 logits")
    if os.path.exists(part):
        os.remove(part)
else:
    n, bs = 10000, 50
    part = os.path.join(HERE, "_10k_partial.npz")
    done = 0
    labels = np.zeros(n, np.int16); conf = np.zeros(n, np.float32); gaps = np.zeros(n, np.float32)
    if os.path.exists(part):
        d = np.load(part); done = int(d["done"]); labels, conf, gaps = d["labels"], d["conf"], d["gap"]
    cfg = O.ClassifyConfig(exact=True)
    t0 = time.time()
    for s in range(done, n, bs):
        l, c, lg, pb = O.classify(model, frames(s, bs), cfg, return_logits=True)
        labels[s:s + bs] = l; conf[s:s + bs] = c; gaps[s:s + bs] = gap_of(pb)
        if (s // bs) % 10 == 9:
            np.savez(part, done=s + bs, labels=labels, conf=conf, gap=gaps)
            print("10k", s + bs, time.time() - t0, flush=True)
    np.savez_compressed(os.path.join(HERE, "r50_exact_10k_noise3.npz"), labels=labels, conf=conf, gap=gaps,
                        blob_sha256=info["sha256"],
                        meta="resnet50 seed1; frames seed 21 ids 0..9999 + gaussian noise sev3 seed 3; single pass; exact")
    if os.path.exists(part):
        os.remove(part)
print("done")
