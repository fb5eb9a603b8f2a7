"""lavender_amd -- MI355X-native (gfx950 / CDNA4) implementation of the LAVENDER pretrain hot path:
Video-Swin encoder + BERT-style fusion encoder + MLM head, forward/backward/optimizer, data-parallel over RCCL.

Drop-in surface (same names as the reference): `LAVENDER_Base`, `EncVideo`, `EncTxt`, `get_vidswin_model`,
`LAVENDER_Pretrain_MLM`, `Agent_Base`, `Agent_Pretrain_MLM`, `WarmupLinearLR`; next-in-line callers of the same path:
`LAVENDER_Pretrain` / `Agent_Pretrain` (task-specific heads) and `LAVENDER_Retrieval_MLM` / `Agent_Retrieval_MLM`.  `VIOLET_Base` is kept as an alias
(the checkpoints are still called ckpt_violet_*, agent.py:176).
"""
from ._lib import LavenderHipError, LIB_PATH  # noqa: F401  (raises ImportError when the HIP library is not built)
from .video_swin import SwinTransformer3D, get_vidswin_model, get_window_size  # noqa: F401
from .model import EncVideo, EncTxt, LAVENDER_Base  # noqa: F401
from .pretrain_mlm import LAVENDER_Pretrain_MLM, Agent_Pretrain_MLM, masking  # noqa: F401
from .pretrain_task_specific import LAVENDER_Pretrain, Agent_Pretrain  # noqa: F401
from .retrieval_mlm import LAVENDER_Retrieval_MLM, Agent_Retrieval_MLM, LAVENDER_RetrievalMlmEval  # noqa: F401
from .captioning import LAVENDER_Captioning  # noqa: F401
from .qa_mlm import (LAVENDER_QAOE_MLM, LAVENDER_QAMC_MLM, LAVENDER_RetMC_MLM, Agent_QAOE_MLM, Agent_QAMC_MLM,  # noqa: F401
                     Agent_RetMC_MLM)
from .agent import Agent_Base, WarmupLinearLR, CrossEntropyIgnore  # noqa: F401

VIOLET_Base = LAVENDER_Base
