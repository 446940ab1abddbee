"""VGGBackBone parameter container (reference: core/modules/net/backbone.py:7-128)."""
from torch import nn

from .vgg import vgg_block


class VGGBackBone(nn.Module):
    def __init__(self, in_channels=1, feat_channels=128, use_batchnorm=False, use_max_pooling=True, padding=1):
        super().__init__()
        if padding not in (0, 1):
            raise AssertionError(padding)  # backbone.py:26
        # (padding=0 with pooling constructs like in the reference; its forward cannot complete there either: see
        # _base.NativeExtractor._pooled_padding0_error)
        self.padding = padding
        self.use_max_pooling = use_max_pooling
        chans = [(in_channels, 64), (64, 64), (64, 128), (128, feat_channels)]
        for i, (ci, co) in enumerate(chans, start=1):
            setattr(self, f"l{i}", nn.Sequential(vgg_block(ci, co, 3, use_batchnorm, padding=padding),
                                                 vgg_block(co, co, 3, use_batchnorm, padding=padding)))

    def layer_blocks(self):
        """[(block, pooled)] in execution order: pool after l1, l2, l3 (backbone.py:116-123)."""
        out = []
        for i in range(1, 5):
            seq = getattr(self, f"l{i}")
            out.append((seq[0], False))
            out.append((seq[1], self.use_max_pooling and i < 4))
        return out

    def forward(self, *a, **k):
        raise RuntimeError("parameter container only; the forward pass is native (see _extract.ExtractorEngine)")
