"""``import ctc_crf`` drop-in: CAT does ``from ctc_crf import CTC_CRF_LOSS as CRFLoss`` and
``from ctc_crf import CRFContext`` (cat/ctc/train.py:118,137).  Put this repository's root on
PYTHONPATH (or `pip install -e .`) and those imports resolve to the MI355X-native implementation."""
from cat_amd.ctc_crf import (CRFContext, CTC_CRF_LOSS, WARP_CTC_LOSS, _CTC_CRF, _CTC_CRF_LOGITS, _WARP_CTC_GPU,  # noqa: F401
                             __version__, ctc_crf_loss)
from cat_amd.ctc_crf import _C  # noqa: F401
