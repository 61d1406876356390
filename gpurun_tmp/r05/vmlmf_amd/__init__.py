"""vmlmf_amd: the VMLMF compressed-LSTM forward/backward hot path as hand-written HIP kernels for
MI355X (gfx950), behind the reference's own nn.Module API.

    from vmlmf_amd import MyVMLMFCell, MyVMLMFCellg2, MyLSTM, Net      # HAR   (models/vmlmf*.py)
    from vmlmf_amd import MyVMLSTM, MyVMLSTMGroup, Model               # LM    (models/vmlmf_lm.py)
    from vmlmf_amd import nll_loss                                     #       (train_test/lm_test.py)
"""
from .cells import MyVMLMFCell, MyVMLMFCellg2, MyVMLMFgCellg2, MyLSTMCell, MyLSTM, Net, TIME_STEPS, RECURRENT_MAX, RECURRENT_MIN
from .lm import MyVMLSTM, MyVMLSTMGroup, Embed, Linear, LSTM, Model
from .functional import (vmlmf_sequence, vmlmf_stack, head_linear, cross_entropy, CrossEntropyLoss, nll_loss, linear_nll, lm_head_loss, embedding, unit_gradient,
                         set_compute_dtype, cache_packed_parameters, dropout, dropout_state, dropout_advance, embedding_dropout)
from . import optim
from .graphed import GraphedTrainStep

__all__ = ["GraphedTrainStep", "vmlmf_stack", "optim", "head_linear", "cross_entropy", "CrossEntropyLoss", "MyVMLMFCell", "MyVMLMFCellg2", "MyVMLMFgCellg2", "MyLSTMCell", "MyLSTM", "Net", "MyVMLSTM", "MyVMLSTMGroup",
           "Embed", "Linear", "LSTM", "Model", "nll_loss", "linear_nll", "lm_head_loss", "embedding", "unit_gradient", "vmlmf_sequence", "dropout", "dropout_state", "dropout_advance",
           "embedding_dropout"]
