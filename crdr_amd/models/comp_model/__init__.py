import os.path as osp

from crdr_amd.utils.misc import import_modules

import_modules("crdr_amd.models.comp_model", osp.dirname(osp.abspath(__file__)), suffix="_model.py")
