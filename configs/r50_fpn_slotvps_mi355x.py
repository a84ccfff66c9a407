# Hot-path configuration of the MI355X build for the R50-FPN Slot-VPS model (Cityscapes-VPS).
# Key names and values of `dynamic_mask_head` / `other_config` follow the reference's
# configs/cityscapes/r50_fpn_slotvps.py:27-107 so that the same dict drives either implementation;
# the reference's own file also loads unchanged through slotvps_amd.config.Config.fromfile.
model = dict(
    type='VPS_Temporal_Slots',
    dynamic_mask_head=dict(
        dh_dim=256,
        num_classes=20,            # 11 stuff + 8 things + no-object
        dim_feedforward=2048,
        nhead=8,
        dropout=0.0,
        activation="gelu",
        dh_num_heads=7,
        per_dh_num_heads=[1, 2, 2, 2],
        feat_num_levels=4,
        merge_operation="concat",
        trans_in_dim=384,
        return_intermediate=True,
        use_focal=True,
        prior_prob=0.01,
        num_cls=2,
        num_reg=2,
        drop_path=0.,
        temporal_query_attention_config=dict(
            d_model=256, dim_feedforward=1024, dropout=0.0, activation="relu", softmax_dim="slots", drop_path=0.),
        apply_temporal_query_atten_stages=[3, 4, 5, 6],
    ),
    other_config=dict(
        proposal_num=100,
        has_no_obj=True,
        pos_config=dict(position_embedding="sine", hidden_dim=256),
    ),
)
# clip geometry of the headline benchmark (BASELINE.json configs[1])
clip = dict(frames=5, height=1024, width=2048)
