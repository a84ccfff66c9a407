# VIPER long-clip configuration (BASELINE.json configs[4]): 1080x1920 frames padded to 1088x1920, T = 10, 200 slots.
# The reference ships no VIPER config or dataset helper (tools/dataset/__init__.py:5-6 commented out); what it has are the
# code branches for num_classes in {23, 24}: 13 stuff classes and image ids encoded as vid * 100000 + fid
# (mmdet/models/detectors/vps_temporal_slots.py:68-70, :220-222; vps_capsule.py:53-57). This file is the synthetic
# configuration SURVEY.md Appendix B describes: 13 stuff + 10 thing classes + no-object = 24 head classes, 23 semantic
# classes, proposal_num = 200; everything else as in the R50 Cityscapes config. Level sizes 34x60, 68x120, 136x240,
# 272x480 (no multiple of the 32-pixel tile except the finest width).
model = dict(
    type='VPS_Temporal_Slots',
    pretrained=None,
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, norm_eval=True,
                  style='pytorch'),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
    panoptic=dict(type='UPSNetFPN', in_channels=256, out_channels=128, num_levels=4, num_things_classes=10,
                  num_classes=23, ignore_label=255, loss_weight=0.5),
    dynamic_mask_head=dict(
        dh_dim=256, num_classes=24, dim_feedforward=2048, nhead=8, dropout=0.0, activation="gelu", dh_num_heads=7,
        per_dh_num_heads=[1, 2, 2, 2], feat_num_levels=4, merge_operation="concat", trans_in_dim=384,
        return_intermediate=True, use_focal=True, prior_prob=0.01, num_cls=2, num_reg=2, drop_path=0.,
        temporal_query_attention_config=dict(
            d_model=256, dim_feedforward=1024, dropout=0.0, activation="relu", softmax_dim="slots", drop_path=0.),
        apply_temporal_query_atten_stages=[3, 4, 5, 6],
    ),
    postprocess_panoptic=dict(
        is_thing_map={i: i > 12 for i in range(24)}, threshold=0.85, fraction_threshold=0.03, pixel_threshold=0.4,
        apply_mask_removal=True, apply_mask_removal_only_ins=True, use_mask_low_constant=False, num_classes=24),
    simple_track_head=dict(num_fcs_query=2, in_channels_query=256, query_matched_weight=1.0),
    other_config=dict(proposal_num=200, has_no_obj=True, pos_config=dict(position_embedding="sine", hidden_dim=256),
                      test_forward_ref_img=True, test_only_save_main_results=True,
                      # MI355X build only: precision mode of the slot head (MultiScaleDynamicMaskHead.MODES); "fp16x2" = the default, the
                      # mode that meets the reference's outputs to 1e-4; "bf16" / "fp16" are opt-in 16-bit storage policies
                      mode="fp16x2"),
)
train_cfg = None
test_cfg = dict(loss_pano_weight=None, class_mapping={i: 12 + i for i in range(1, 11)})
clip = dict(frames=10, height=1088, width=1920)
