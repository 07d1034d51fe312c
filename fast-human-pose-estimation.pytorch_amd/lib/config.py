"""`config.cfg` / `update_config` with the reference's behaviour (/root/reference/lib/config/default.py:18-164):
defaults <- YAML file (--cfg) <- `KEY VALUE` list from the command line.  yacs is not available here, so this is a
small attribute+item-access dict over PyYAML (HRNet reads both cfg.MODEL.EXTRA and cfg['MODEL']['EXTRA'])."""
import ast
import copy

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def defrost(self):
        pass

    def freeze(self):
        pass

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_dict(v)
            else:
                self[k] = _wrap(v)

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, 'opts must be KEY VALUE pairs'
        for k, v in zip(opts[0::2], opts[1::2]):
            node = self
            parts = k.split('.')
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            try:
                v = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                pass
            node[parts[-1]] = _wrap(v)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        n = CfgNode()
        for k, x in v.items():
            n[k] = _wrap(x)
        return n
    if isinstance(v, str):                      # yacs literal-evals strings such as '(0,)'
        try:
            lit = ast.literal_eval(v)
            if isinstance(lit, (tuple, list)):
                return lit
        except (ValueError, SyntaxError):
            pass
    return v


def _defaults():
    return _wrap({
        'OUTPUT_DIR': 'output', 'LOG_DIR': 'log', 'DATA_DIR': '', 'GPUS': (0,), 'WORKERS': 4, 'PRINT_FREQ': 20,
        'AUTO_RESUME': False, 'PIN_MEMORY': True, 'RANK': 0,
        'CUDNN': {'BENCHMARK': True, 'DETERMINISTIC': False, 'ENABLED': True},
        'MODEL': {'NAME': 'hourglass', 'INIT_WEIGHTS': False, 'PRETRAINED': '', 'NUM_JOINTS': 16, 'TAG_PER_JOINT': True,
                  'TARGET_TYPE': 'gaussian', 'IMAGE_SIZE': [256, 256], 'HEATMAP_SIZE': [64, 64], 'SIGMA': 2, 'DTYPE': 'fp32',
                  'EXTRA': {'NUM_FEATURES': 256, 'NUM_STACKS': 8, 'NUM_BLOCKS': 1}},
        'LOSS': {'USE_OHKM': False, 'TOPK': 8, 'USE_TARGET_WEIGHT': True, 'USE_DIFFERENT_JOINTS_WEIGHT': False},
        'DATASET': {'ROOT': '', 'DATASET': 'synthetic', 'TRAIN_SET': 'train', 'TEST_SET': 'valid', 'NUM_SAMPLES': 256,
                    'NUM_VALID_SAMPLES': 64},
        'TRAIN': {'LR_FACTOR': 0.1, 'LR_STEP': [90, 120], 'LR': 0.00025, 'OPTIMIZER': 'adam', 'MOMENTUM': 0.9, 'WD': 0.0001,
                  'NESTEROV': False, 'GAMMA1': 0.99, 'GAMMA2': 0.0, 'BEGIN_EPOCH': 0, 'END_EPOCH': 140, 'RESUME': False,
                  'CHECKPOINT': '', 'BATCH_SIZE_PER_GPU': 32, 'SHUFFLE': True},
        'TEST': {'BATCH_SIZE_PER_GPU': 32, 'FLIP_TEST': False, 'POST_PROCESS': False, 'SHIFT_HEATMAP': False, 'MODEL_FILE': ''},
        'KD': {'TRAIN_TYPE': 'NORMAL', 'TEACHER': '', 'ALPHA': 0.5},      # default.py:122-126
        'DEBUG': {'DEBUG': False, 'SAVE_BATCH_IMAGES_GT': False, 'SAVE_BATCH_IMAGES_PRED': False,
                  'SAVE_HEATMAPS_GT': False, 'SAVE_HEATMAPS_PRED': False},
    })


cfg = _defaults()


def update_config(cfg, args):
    """default.py:137-164: file first, then the KEY VALUE list."""
    if getattr(args, 'cfg', None):
        cfg.merge_from_file(args.cfg)
    if getattr(args, 'opts', None):
        cfg.merge_from_list(args.opts)
    for k in ('modelDir', 'logDir', 'dataDir'):
        v = getattr(args, k, '')
        if v:
            cfg[{'modelDir': 'OUTPUT_DIR', 'logDir': 'LOG_DIR', 'dataDir': 'DATA_DIR'}[k]] = v
    return cfg
