"""-m gpu: the widening rows of SURVEY.md 8(f) added after the last GPU session of round 1 -- the same cases the emulator
tests run (tests/test_transforms_emu.py, tests/test_e2e_emu.py), here through the real .so on the MI355X.  Kept in a file that
sorts after the others so that `pytest -x` reaches every test that has already run on the hardware first."""
import pytest

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_preproc_vs_reference():
    import preproc_cases as pc

    print("arrays", pc.case_preproc_vs_reference(DEV))
    print("boxes", pc.case_bbox_large(DEV))


def test_preproc_api():
    import preproc_cases as pc

    pc.case_preproc_api(DEV)


def test_dynunet_vs_reference():
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_vs_reference(DEV))
    dc.case_dynunet_api(DEV)


def test_dynunet_sliding_window_vs_reference():
    import dynunet_cases as dc

    print("max |dlogit|", dc.case_dynunet_sliding_window(DEV))
