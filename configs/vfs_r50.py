# VFS ResNet-50 (frame-level similarity, one frame per view) -- model / schedule settings
# equal to the reference's r50_nc_sgd_cos_100e_r5_1xNx2_k400 config (model, train_cfg, test_cfg,
# optimizer, img_norm_cfg, train_pipeline).  Decoding steps are listed as in the reference but run outside this
# package; vfs_amd.pipeline.GpuTrainPipeline consumes the augmentation / formatting steps.
_norm = dict(type='SyncBN', requires_grad=True)
model = dict(
    type='SimSiamBaseTracker',
    backbone=dict(type='ResNet', depth=50, pretrained=None, out_indices=(3,), norm_cfg=_norm,
                  norm_eval=False, zero_init_residual=True),
    img_head=dict(type='SimSiamHead', in_channels=2048, norm_cfg=dict(type='SyncBN'),
                  num_projection_fcs=3, projection_mid_channels=2048, projection_out_channels=2048,
                  num_predictor_fcs=2, predictor_mid_channels=512, predictor_out_channels=2048,
                  with_norm=True, loss_feat=dict(type='CosineSimLoss', negative=False),
                  spatial_type='avg'))
train_cfg = dict(intra_video=False)
test_cfg = dict(precede_frames=20, topk=10, temperature=0.07, strides=(1, 2, 1, 1), out_indices=(2,),
                neighbor_range=36, with_first=True, with_first_neighbor=True, output_dir='eval_results')
# 2 clips x 1 frame per video, 32 videos per GPU
clip_len, num_clips, videos_per_gpu = 1, 2, 32
optimizer = dict(type='SGD', lr=0.05, momentum=0.9, weight_decay=0.0001)
lr_config = dict(policy='CosineAnnealing', min_lr=0, by_epoch=False)
total_epochs = 100
dist_params = dict(backend='nccl')
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)
train_pipeline = [
    dict(type='DecordInit'),
    dict(type='SampleFrames', clip_len=1, frame_interval=0, num_clips=2, out_of_bound_opt='loop'),
    dict(type='DecordDecode'),
    dict(type='RandomResizedCrop', area_range=(0.2, 1.), same_across_clip=False, same_on_clip=False),
    dict(type='Resize', scale=(224, 224), keep_ratio=False),
    dict(type='Flip', flip_ratio=0.5, same_across_clip=False, same_on_clip=False),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='FormatShape', input_format='NCTHW'),
    dict(type='Collect', keys=['imgs', 'label'], meta_keys=[]),
    dict(type='ToTensor', keys=['imgs', 'label']),
]
