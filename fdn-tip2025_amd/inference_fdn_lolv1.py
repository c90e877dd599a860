"""LOL-v1 inference driver on the HIP path: the role of the reference's inference_fdn_lolv1.py:1-66 (FDN_lolv1, dim 24, with
LPNet_lolv1; ratio_i = mean(Grayscale(padded frame)) / LPNet(padded frame), :57-61), paths as arguments, per-image host work on
the GPU, frames of equal size batched.  LOL-v1 frames are 400 x 600 -> reflect-padded to 416 x 608, a shape with compile-time FFT
plans (rows 19 x 16 / 19 x 8, columns 13 x 32 / 16 / 8).  Needs a ROCm GPU and the built libfdn_hip.so; there is no CPU fallback.

    python inference_fdn_lolv1.py --fdn FDN_lolv1.pth --lpnet LPNet_lolv1.pth --input 'testlow/*.png' --output out/
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from inference_fdn_lolblur import run_driver  # noqa: E402


def main():
    def build():
        from basicsr.models.archs.fdnlol24_arch import FDN_lolv1
        from basicsr.models.archs.LPNet_arch import I_predict_net
        return FDN_lolv1(), I_predict_net()
    run_driver(__doc__, build, ratio_mode="lolv1", fdn_keys="FDN_lolv1 checkpoint ({'params': state_dict}, 1503 keys)")


if __name__ == "__main__":
    main()
