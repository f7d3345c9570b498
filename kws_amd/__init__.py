"""kws_amd -- MI355X-native FastGRNN recurrent cell behind the reference's
``fastgrnn_cuda`` / ``FastGRNNCUDA`` operator boundary (adithom/KWS rnn.py:738-972,
cuda/fastgrnn_cuda.cpp:235-240).  HIP kernels + C ABI live in ``kws_amd/csrc``;
there is no CPU path in this package."""
from . import _lib  # noqa: F401
from . import fastgrnn_cuda  # noqa: F401
from . import utils  # noqa: F401
from . import head  # noqa: F401
from .head import KeywordHead, keyword_loss  # noqa: F401
from .rnn import (FastGRNNCUDA, FastGRNNCUDACell, FastGRNNFunction,  # noqa: F401
                  FastGRNNUnrollFunction)
from .model import RNNClassifierModel  # noqa: F401
from .graph import GraphedStep  # noqa: F401

__all__ = ["fastgrnn_cuda", "utils", "head", "FastGRNNCUDA", "FastGRNNCUDACell", "FastGRNNFunction",
           "FastGRNNUnrollFunction", "KeywordHead", "keyword_loss", "RNNClassifierModel", "GraphedStep"]
