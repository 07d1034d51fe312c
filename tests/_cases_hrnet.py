"""Shared definition of the small HRNet FPD case pinned by tests/golden/hrnet_tiny.npz (student: widths 8..64, one block
per branch; teacher: widths 16..128; the W32/W48 topology of experiments/fpd_coco/hrnet/*.yaml scaled down)."""

CONFIG = dict(s=dict(widths=[8, 16, 32, 64], blocks=1, modules=(1, 2, 1)),
              t=dict(widths=[16, 32, 64, 128], blocks=1, modules=(1, 1, 1)),
              joints=5, batch=2, image=(96, 128), heat=(24, 32), alpha=0.5, grad_stride=97)


def extra_cfg(m):
    w, b, mods = m['widths'], m['blocks'], m['modules']
    return {'FINAL_CONV_KERNEL': 1,
            'STAGE2': dict(NUM_MODULES=mods[0], NUM_BRANCHES=2, BLOCK='BASIC', NUM_BLOCKS=[b] * 2, NUM_CHANNELS=w[:2], FUSE_METHOD='SUM'),
            'STAGE3': dict(NUM_MODULES=mods[1], NUM_BRANCHES=3, BLOCK='BASIC', NUM_BLOCKS=[b] * 3, NUM_CHANNELS=w[:3], FUSE_METHOD='SUM'),
            'STAGE4': dict(NUM_MODULES=mods[2], NUM_BRANCHES=4, BLOCK='BASIC', NUM_BLOCKS=[b] * 4, NUM_CHANNELS=w[:4], FUSE_METHOD='SUM')}
