"""(B,H) merge/split helpers — counterpart of /root/reference/src/common/utils.py:3-21."""


def merge_bh(x):
    if x.dim() == 3:
        return x, None
    b, h, n, d = x.shape
    return x.reshape(b * h, n, d), (b, h)


def split_bh(x, bh_shape):
    if bh_shape is None:
        return x
    b, h = bh_shape
    _, n, d = x.shape
    return x.reshape(b, h, n, d)


def split_bh_lse(lse, bh_shape):
    if bh_shape is None:
        return lse
    b, h = bh_shape
    _, n = lse.shape
    return lse.reshape(b, h, n)
