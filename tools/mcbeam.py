#!/usr/bin/env python3
"""mcbeam -- beamform a multichannel WAV file on the GPU and print the DOA per frame.

The counterpart of the reference's CLI (src/programs/mcabeamf.cpp: libsndfile in -> de-interleave ->
SourceSeparationAndLocalisation(sampleRate, 4-mic array, 1 source, usePowerFloor=false) -> mono out + DOA text),
on top of the Python mirror of the module API.  16-bit PCM RIFF/WAV in and out (the `wave` module; no libsndfile).

  python tools/mcbeam.py -i in.wav -o out.wav [-d doa.txt] [--mics 0,0.07,0.175,0.21] [--sources 1] [--step 5]
  two-channel input with --binaural runs FreqGCCBinauralLocalisation (3 degree grid) instead.
"""
import argparse
import os
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import api  # noqa: E402


def read_wav(path):
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise SystemExit("only 16-bit PCM WAV is supported")
        fs, nch, n = w.getframerate(), w.getnchannels(), w.getnframes()
        x = np.frombuffer(w.readframes(n), dtype="<i2").reshape(-1, nch).T      # de-interleave (mcabeamf.cpp:101-110)
    return fs, (x.astype(np.float32) / 32768.0)


def write_wav(path, fs, x):
    y = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(y.shape[0]); w.setsampwidth(2); w.setframerate(fs)
        w.writeframes(y.T.tobytes())


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--input", required=True)
    ap.add_argument("-o", "--output")
    ap.add_argument("-d", "--doa-file")
    ap.add_argument("--mics", default="-2.25,-1.25,1.25,2.25", help="x coordinates in metres (mcabeamf.cpp:182 default)")
    ap.add_argument("--sources", type=int, default=1)
    ap.add_argument("--step", type=float, default=5.0, help="DOA grid step in degrees")
    ap.add_argument("--power-floor", action="store_true", help="usePowerFloor=true (the CLI of the reference passes false, :194)")
    ap.add_argument("--binaural", action="store_true")
    a = ap.parse_args()
    fs, x = read_wav(a.input)
    xs = [float(v) for v in a.mics.split(",")]
    # frame length from the sample rate, like the reference's modules (N = 2^order, hop N/2)
    seconds = api.FreqGCCBinauralLocalisation.FRAME_SECONDS if a.binaural else api.SourceSeparationAndLocalisation.FRAME_SECONDS
    hop = (1 << api.calculate_order_from_sample_rate(fs, seconds)) // 2
    F = x.shape[1] // hop - 1
    if F < 1:
        raise SystemExit("input shorter than one %d-sample frame" % (2 * hop))
    x = np.ascontiguousarray(x[:, :(F + 1) * hop])
    out = sys.stdout if not a.doa_file else open(a.doa_file, "w")
    if a.binaural:
        if x.shape[0] != 2 or len(xs) != 2:
            raise SystemExit("--binaural needs a 2-channel file and two microphone positions")
        loc = api.FreqGCCBinauralLocalisation(fs, xs, a.power_floor, 3.0 if a.step == 5.0 else a.step)
        r = loc.process(x)
        for t in range(F):
            if "voiced" in r and not r["voiced"][0, t]:
                continue                                   # gated out: the reference reports nothing for the frame
            out.write("%d %.3f %.4f\n" % (t, np.rad2deg(r["doa"][0, t]), r["prob"][0, t]))
        return
    if x.shape[0] != len(xs):
        raise SystemExit("the file has %d channels but %d microphone positions were given" % (x.shape[0], len(xs)))
    ssl = api.SourceSeparationAndLocalisation(fs, xs, a.sources, a.power_floor, a.step)
    ssl.set_callback(lambda doa, prob, power, n: out.write(" ".join("%.3f %.4f" % (doa[s], prob[s]) for s in range(n)) + "\n"))
    audio, _ = ssl.process(x)
    if a.output:
        write_wav(a.output, fs, audio[:a.sources])


if __name__ == "__main__":
    main()
