"""`tester.py` of the MI355X path (mirror of the reference's tester.py:44-141): restore a checkpoint, then either
evaluate the test split (`test`, result file next to the checkpoint) or time the scoring pass (`--latency`).

    python -m legommenders_amd.tester --data config/data/synthetic.yaml --model config/model/naml.yaml \
        --embed config/embed/glove.yaml --hidden_size 256 --batch_size 64 --load_sign <signature> [--latency --num_batches 1000]

What `--latency` times: the reference brackets `legommender(batch=batch)` of every test batch with
`Env.latency_timer` (base_lego.py:373-381) -- with `use_fast_eval` that is the id-gather of the cached user / item
vectors plus the dot (model/legommender.py:153-157,202-214,282); the caches are built before the loop and are not
part of the figure.  The same bracket is taken here around `Evaluator.scores` on `batch_size` test rows per step,
with a device synchronise on both sides (the reference's wall clock on a CUDA device lacks it and so measures the
launch only), for `num_batches` steps (StatusTimer.total_count -> StopIteration, utils/timer.py:61-78)."""
from __future__ import annotations

import os
import time

import torch

from legommenders_amd.config_init import CommandInit, Obj
from legommenders_amd.trainer import Trainer


class StatusTimer:
    """utils/timer.py:40-95: toggle timer of one status, mean in ms, StopIteration once `total_count` runs completed"""

    def __init__(self, total_count=0):
        self.total_time, self.start_time, self.timing, self.count, self.total_count = 0.0, None, False, 0, total_count

    def run(self):
        now = time.time()
        if not self.timing:
            self.timing, self.start_time = True, now
            return
        self.total_time += now - self.start_time
        self.timing = False
        self.count += 1
        if self.total_count and self.count >= self.total_count:
            raise StopIteration

    def avgms(self):
        return self.total_time / self.count * 1000 if self.count else 0.0


class Tester(Trainer):
    def test(self):
        res = super().test()
        if self.rank == 0:
            lines = [f"{k}: {v:.4f}" for k, v in res.items()]            # tester.py:69-74
            for line in lines:
                self.log(line)
            with open(os.path.join(self.ckpt_dir, self.signature + ".result"), "w") as f:
                f.write("\n".join(lines))
        return res

    def latency(self):
        rows = self.world["test"]
        B = self.B
        st = StatusTimer(total_count=int(self.config.num_batches or 0))
        self.evaluator.build_caches()
        users, items = torch.as_tensor(rows["user"]), torch.as_tensor(rows["item"])
        dev = self.device
        users, items = users.to(dev, torch.int32), items.to(dev, torch.int32)
        n = users.numel()
        try:
            # one pass over the test rows, cut short by the timer (tester.py:99-103)
            for s in range(0, n, B):
                torch.cuda.synchronize()
                st.run()
                self.evaluator.scores(users[s:s + B], items[s:s + B])
                torch.cuda.synchronize()
                st.run()
        except (KeyboardInterrupt, StopIteration):
            pass
        self.log(f"Total {st.count} steps, avg ms {st.avgms():.4f}")
        return st

    def run(self):
        if self.config.latency:
            return self.latency()
        return self.test()


def get_configurations(kwargs=None) -> Obj:
    return CommandInit(
        required_args=["data", "model"],
        default_args=dict(embed="config/embed/null.yaml", exp="config/exp/default.yaml", hidden_size=256,
                          item_hidden_size="${hidden_size}$", latency=False, num_batches=1000),
    ).parse(kwargs=kwargs)


if __name__ == "__main__":
    Tester(config=get_configurations()).run()
