import os.path as osp

from crdr_amd.utils.misc import import_modules

import_modules("crdr_amd.models.subnet.hyperprior", osp.dirname(osp.abspath(__file__)), suffix="_hyperprior.py")
