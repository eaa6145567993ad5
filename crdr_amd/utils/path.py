"""Job-directory / checkpoint naming (same file layout as src/utils/path.py:13-47):
<ckpt_root>/<exp>/{model,sample}/, model/<label>_iter<N or NK>.pth.tar, log_loss.csv, eval_result.csv."""
from __future__ import annotations

import os
import os.path as osp
from datetime import datetime
from typing import Dict


def check_file_exist(filename: str, msg_tmpl: str = 'file "{}" does not exist') -> None:
    if not osp.isfile(filename):
        raise FileNotFoundError(msg_tmpl.format(filename))


class PathHandler:
    def __init__(self, ckpt_root: str, exp: str) -> None:
        self.ckpt_root, self.exp = ckpt_root, exp
        self.job_dir = osp.join(ckpt_root, exp)

    def make_job_dir(self) -> None:
        for sub in ("model", "sample"):
            os.makedirs(osp.join(self.job_dir, sub), exist_ok=True)

    def get_exp_path_dict(self) -> Dict[str, str]:
        stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        jd = self.job_dir
        return dict(ckpt_root=self.ckpt_root, job_dir=jd, model_dir=osp.join(jd, "model"),
                    sample_dir=osp.join(jd, "sample"), log_loss_path=osp.join(jd, "log_loss.csv"),
                    log_eval_path=osp.join(jd, "eval_result.csv"), log_msg_path=osp.join(jd, f"train_{stamp}.log"))

    @staticmethod
    def iter2str(itr: int) -> str:
        return f"{itr // 1000}K" if itr % 1000 == 0 else str(itr)

    def get_ckpt_path(self, label: str, itr: int) -> str:
        return osp.join(self.job_dir, "model", f"{label}_iter{self.iter2str(itr)}.pth.tar")
