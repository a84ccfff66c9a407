# Swin-L-FPN Slot-VPS (BASELINE.json configs[3]) for the MI355X build. Same keys as the reference's
# configs/cityscapes/swinL_fpn_slotvps.py:2-112 (which also loads unchanged through slotvps_amd.config.Config.fromfile
# where the reference tree exists); this copy makes the model buildable where it does not (the GPU box).
# Differences from the R50 config: Swin-L backbone (embed 192, depths 2/2/18/2, heads 6/12/24/48, window 7), FPN inputs
# [192, 384, 768, 1536], head activation relu / temporal activation gelu (:41, :58), no drop_path key in the head.
model = dict(
    type='VPS_Temporal_Slots',
    pretrained=None,
    backbone=dict(type='SwinTransformer', embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7,
                  mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.5,
                  ape=False, patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False),
    neck=dict(type='FPN', in_channels=[192, 384, 768, 1536], out_channels=256, num_outs=5),
    panoptic=dict(type='UPSNetFPN', in_channels=256, out_channels=128, num_levels=4, num_things_classes=8,
                  num_classes=19, ignore_label=255, loss_weight=0.5),
    dynamic_mask_head=dict(
        dh_dim=256, num_classes=20, dim_feedforward=2048, nhead=8, dropout=0.0, activation="relu", dh_num_heads=7,
        per_dh_num_heads=[1, 2, 2, 2], feat_num_levels=4, merge_operation="concat", trans_in_dim=384,
        return_intermediate=True, use_focal=True, prior_prob=0.01, num_cls=2, num_reg=2,
        temporal_query_attention_config=dict(
            d_model=256, dim_feedforward=1024, dropout=0.0, activation="gelu", softmax_dim="slots", drop_path=0.),
        apply_temporal_query_atten_stages=[3, 4, 5, 6],
    ),
    postprocess_panoptic=dict(
        is_thing_map={i: i > 10 for i in range(20)}, threshold=0.85, fraction_threshold=0.03, pixel_threshold=0.4,
        apply_mask_removal=True, apply_mask_removal_only_ins=True, use_mask_low_constant=False),
    simple_track_head=dict(num_fcs_query=2, in_channels_query=256, query_matched_weight=1.0),
    other_config=dict(proposal_num=100, has_no_obj=True, pos_config=dict(position_embedding="sine", hidden_dim=256),
                      test_forward_ref_img=True, test_only_save_main_results=True,
                      # MI355X build only: precision mode of the slot head (MultiScaleDynamicMaskHead.MODES); "fp16x2" = the default, the
                      # mode that meets the reference's outputs to 1e-4; "bf16" / "fp16" are opt-in 16-bit storage policies
                      mode="fp16x2"),
)
train_cfg = None
test_cfg = dict(loss_pano_weight=None, class_mapping={1: 11, 2: 12, 3: 13, 4: 14, 5: 15, 6: 16, 7: 17, 8: 18})
clip = dict(frames=5, height=1024, width=2048)
