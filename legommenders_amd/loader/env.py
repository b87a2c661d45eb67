"""Process-global run-time flags (mirror of the reference's loader/env.py:4-61)."""
import torch


class Env:
    device = None                 # torch.device of the MI355X this process drives
    simple_dev = False
    UNSET = -1                    # pad id (loader/env.py:10)
    is_training = True
    is_evaluating = False
    is_testing = False
    item_cache = False
    user_cache = False
    lm_cache = False
    data_name = None              # Env.ph.data_name of the reference: names the on-disk layer cache directory

    @classmethod
    def train(cls):
        cls.is_training, cls.is_evaluating, cls.is_testing = True, False, False

    @classmethod
    def dev(cls):
        cls.is_training, cls.is_evaluating, cls.is_testing = False, True, False

    @classmethod
    def test(cls):
        cls.is_training, cls.is_evaluating, cls.is_testing = False, False, True

    @classmethod
    def set_device(cls, device):
        """`--cuda idx` -> cuda:idx.  `--cuda -1` (the reference's CPU path, base_lego.py:281-296) is refused:
        this package is the MI355X path only and has no CPU fallback."""
        if device is None or str(device) in ("-1", "cpu"):
            from legommenders_amd._lib import LegoHipError
            raise LegoHipError("--cuda -1 / cpu requested: the MI355X-native path has no CPU fallback; "
                               "run the reference for CPU execution")
        cls.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        return cls.device

    @classmethod
    def set_item_cache(cls, v):
        cls.item_cache = v

    @classmethod
    def set_user_cache(cls, v):
        cls.user_cache = v

    @classmethod
    def set_lm_cache(cls, v):
        cls.lm_cache = v
