"""GPU: the reference encoder with its whole RD search replaced by hop_encode_frame writes the reference's bitstream.

oracle/_ref/TAppEncoderPic is the reference encoder application, built in the build container from the reference's sources by oracle/Makefile.ref, with ONE member replaced
(oracle/enc_shim_pic.cpp, the picture-level binding of INTEGRATION.md section 4): TEncCu::compressCU.  A picture goes through libhophip.so -- hop_ctx_create, hop_upload_orig,
hop_encode_frame, hop_levels_download, hop_recon_download, hop_rd_fraction_download -- and each CTU's TComDataCU is filled from what came back.  The counting pass, the
entropy coder of encodeSlice, deblocking, SAO, the picture hash and the NAL writer are the reference's own object code; nothing of the CPU restatement is linked into the
program.  The bitstream and the reconstruction it writes must be byte-identical to those of the UNMODIFIED reference encoder (tests/golden/encoder_hop_pic.json, made in the
build container by oracle/make_golden21.py): raster order and WaveFrontSynchro (coded as a wavefront with the rows' requests batched), pictures with partial CTUs on the
right and below, two pictures through one context.  A wrong level, flag, vector or reconstructed sample anywhere changes the md5."""
import os

import pytest

from hoputil import PIC_CASES, ROOT
from test_encoder_pic import check, run_binding

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "oracle", "_ref", "TAppEncoderPic")


@pytest.mark.parametrize("key", list(PIC_CASES))
def test_reference_encoder_over_the_library_writes_the_reference_bitstream(key):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/TAppEncoderPic was not built (it is built where the reference tree is present and travels with the snapshot)")
    got, counts = run_binding(EXE, key, {})
    check(key, got, counts)


@pytest.mark.parametrize("key", ["200x104_raster", "448x192_wpp", "128x64_2frames"])
def test_reference_encoder_with_the_library_deblocking_too(key):
    """HOP_PIC_DEBLOCK: TComLoopFilter::loopFilterPic replaced as well, by hop_deblock_frame on the reconstruction the context holds; SAO, the hash and the bitstream are the
    reference's and see the library's deblocked picture: same md5s"""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/TAppEncoderPic was not built")
    got, counts = run_binding(EXE, key, {"HOP_PIC_DEBLOCK": "1"})
    assert counts["deblocked"] == PIC_CASES[key]["frames"]
    check(key, got, counts)


@pytest.mark.parametrize("key", ["200x104_raster", "448x192_wpp", "128x64_2frames", "192x128_raster", "136x72_plain8", "136x72_plain10"])
def test_reference_encoder_with_the_library_deblocking_and_sao(key):
    """HOP_PIC_DEBLOCK + HOP_PIC_SAO: everything between the original and the entropy coder is the library's -- hop_encode_frame, hop_deblock_frame, hop_sao_frame (its
    statistics and offsetting kernels, its decision starting from hop_rd_fraction_download's fraction); the reference writes the SAO parameters the library chose and hashes
    the library's final picture: same md5s as the unmodified encoder"""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/TAppEncoderPic was not built")
    got, counts = run_binding(EXE, key, {"HOP_PIC_DEBLOCK": "1", "HOP_PIC_SAO": "1"})
    assert counts["deblocked"] == PIC_CASES[key]["frames"] and counts["sao"] == PIC_CASES[key]["frames"]
    check(key, got, counts)


DEC = os.path.join(ROOT, "oracle", "_ref", "TAppDecoderAbi")


@pytest.mark.parametrize("key", ["192x128_raster", "200x104_raster", "448x192_wpp", "128x64_2frames"])
def test_reference_decoder_over_the_library_decodes_to_the_reconstruction(key):
    """SURVEY 8(f)-4, the decoder side of the shared predictor: oracle/_ref/TAppDecoderAbi is the reference DECODER with TComPrediction::xPredInterLumaBlk / ChromaBlk
    replaced by hop_pred_inter and TDecCu::xFindSSRef2Copy followed by hop_ssref_commit_cus (oracle/dec_shim_abi.cpp): every SS / GT prediction of the decode comes from
    the SS reference resident on the device.  It decodes the stream the encoder-side binding has just written (itself the reference's stream, byte for byte) to exactly the
    encoder's reconstruction, and the decoder's own check of the picture hash SEI passes."""
    if not (os.path.exists(EXE) and os.path.exists(DEC)):
        pytest.skip("oracle/_ref/TAppEncoderPic / TAppDecoderAbi were not built")
    got, counts = run_binding(EXE, key, {"HOP_PIC_DEBLOCK": "1", "HOP_PIC_SAO": "1"}, decoder=DEC)
    check(key, got, counts)
    assert counts["dec_pictures"] == PIC_CASES[key]["frames"] and counts["dec_predictions"] > 20 and counts["dec_gt"] > 0 and counts["dec_commits"] > 10, counts
