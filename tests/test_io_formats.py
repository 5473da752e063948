"""the on-disk formats either side of the path, against golden vectors produced by the reference's own
readers / writers (tests/golden/make_golden_formats.py -> ref_formats.npz) and through round trips.  CPU only."""
import os

import numpy as np
import pytest
import torch

from das3r_amd import io_formats as io

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_formats.npz"))


def test_quaternion_conversions_match_reference():
    for q, R, back in zip(G["quat_in"], G["quat_R"], G["quat_back"]):
        assert np.allclose(io.qvec2rotmat(q), R, atol=1e-12)
        assert np.allclose(io.rotmat2qvec(R), back, atol=1e-10)
    assert np.allclose(io.matrix_to_quat_wxyz(G["quat_R"]), G["r2q_out"], atol=2e-6)          # reference ran in fp32
    assert np.allclose(np.stack([io.rotation_to_quat_wxyz(R) for R in G["quat_R"]]), G["rtq_out"], atol=1e-12)


def test_tumpose_to_c2w_reproduces_reference_including_its_quaternion_order():
    got = np.stack([io.tumpose_to_c2w(p) for p in G["tum_in"]])
    assert np.allclose(got, G["tum_c2w"], atol=2e-6)   # the reference builds the rotation in fp32 torch
    # the quirk: NOT the rotation of the (w, x, y, z) quaternion stored in the file
    assert not np.allclose(got[0][:3, :3], io.qvec2rotmat(G["tum_in"][0][3:]), atol=1e-3)


def test_colmap_text_writers_and_readers(tmp_path):
    cam_p, img_p = tmp_path / "cameras.txt", tmp_path / "images.txt"
    names = [f"frame_{i:04d}.png" for i in range(5)]
    io.write_colmap_cameras_text(cam_p, (512, 208), G["fmt_K"])
    io.write_colmap_images_text(img_p, list(G["tum_c2w"][:5]), names)
    assert cam_p.read_bytes() == G["cam_txt"].tobytes(), "cameras.txt must equal the reference writer's output byte for byte"
    assert img_p.read_bytes() == G["img_txt"].tobytes(), "images.txt must equal the reference writer's output byte for byte"
    cams, imgs = io.read_colmap_cameras_text(cam_p), io.read_colmap_images_text(img_p)
    assert sorted(cams) == [1, 2, 3, 4, 5] and sorted(imgs) == [1, 2, 3, 4, 5]
    assert np.array_equal(np.array([[cams[i]["width"], cams[i]["height"]] for i in sorted(cams)]), G["parsed_cam_wh"])
    assert np.array_equal(np.stack([cams[i]["params"] for i in sorted(cams)]), G["parsed_cam_params"])
    assert np.array_equal(np.stack([imgs[i]["qvec"] for i in sorted(imgs)]), G["parsed_img_qvec"])
    assert np.array_equal(np.stack([imgs[i]["tvec"] for i in sorted(imgs)]), G["parsed_img_tvec"])
    assert [imgs[i]["name"] for i in sorted(imgs)] == names and [imgs[i]["camera_id"] for i in sorted(imgs)] == list(G["parsed_img_camid"])
    cam_p.write_text("# comment\n1 SIMPLE_RADIAL 4 4 1 2 2 0\n")
    with pytest.raises(ValueError):
        io.read_colmap_cameras_text(cam_p)


def test_depth_to_points_matches_reference():
    from das3r_amd.model import depth_to_points
    pts = depth_to_points(torch.from_numpy(G["d2p_K"]), torch.from_numpy(G["d2p_pose"]), torch.from_numpy(G["d2p_depth"]))
    assert np.allclose(pts.reshape(3, -1, 3).numpy(), G["d2p_pts"], atol=2e-5)


def test_tum_and_intrinsics_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    t, xyz = np.arange(4, dtype=np.float64), rng.normal(size=(4, 3))
    q = rng.normal(size=(4, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    p = tmp_path / "pred_traj.txt"
    io.write_tum_trajectory(p, t, xyz, q)
    t2, xyz2, q2 = io.read_tum_trajectory(p)
    assert np.array_equal(t, t2) and np.array_equal(xyz, xyz2) and np.array_equal(q, q2)
    assert len(p.read_text().splitlines()[0].split()) == 8
    K = rng.uniform(1, 700, size=(4, 3, 3)).astype(np.float32)
    np.savetxt(tmp_path / "pred_intrinsics.txt", K.reshape(4, 9))
    assert np.allclose(io.read_pred_intrinsics(tmp_path / "pred_intrinsics.txt"), K, rtol=1e-6)


def test_ply_layout_and_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    P = 37
    xyz, f_dc, f_rest = torch.randn(P, 3, generator=g), torch.randn(P, 1, 3, generator=g), torch.randn(P, 15, 3, generator=g)
    opa, sc, rot = torch.randn(P, 1, generator=g), torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g)
    conf = torch.rand(P, 1, generator=g) * 0.9 + 0.05
    path = tmp_path / "point_cloud" / "iteration_4000" / "point_cloud.ply"
    io.save_gaussians_ply(path, xyz, f_dc, f_rest, opa, sc, rot, conf)
    raw = path.read_bytes()
    head = raw[:raw.index(b"end_header\n") + 11].decode("ascii").splitlines()
    assert head[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    names = [ln.split()[2] for ln in head if ln.startswith("property")]
    assert names == io.ply_attribute_names(3, 45) and all(ln.split()[1] == "float" for ln in head if ln.startswith("property"))
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[6 + 48:6 + 51] == ["opacity_ori", "opacity", "conf_static"]
    assert len(raw) == raw.index(b"end_header\n") + 11 + P * len(names) * 4
    d = io.load_gaussians_ply(path, max_sh_degree=3)
    assert np.array_equal(d["xyz"], xyz.numpy()) and np.array_equal(d["features_dc"], f_dc.numpy())
    assert np.array_equal(d["features_rest"], f_rest.numpy()) and np.array_equal(d["opacity"], opa.numpy())
    assert np.array_equal(d["scaling"], sc.numpy()) and np.array_equal(d["rotation"], rot.numpy())
    assert np.array_equal(d["conf_static"], conf.numpy())
    o = torch.sigmoid(opa) * conf
    assert np.allclose(d["opacity_with_conf"], torch.log(o / (1 - o)).numpy(), atol=1e-5)
    v = io.read_ply_vertices(path)
    assert np.array_equal(v["f_rest_15"], f_rest[:, 0, 1].numpy()), "features are stored channel-major (transpose(1, 2).flatten)"
    assert float(np.abs(v["nx"]).max()) == 0.0
    with pytest.raises(ValueError):
        io.load_gaussians_ply(path, max_sh_degree=2)


def test_load_sequence_from_a_directory(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(2)
    F, H, W = 3, 8, 12
    seq = tmp_path / "market_2"
    for sub in ("images", "sparse/0", "depth_maps", "confidence_maps", "dyna_avg", "dynamic_masks"):
        os.makedirs(seq / sub)
    q = rng.normal(size=(F, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    xyz = rng.normal(size=(F, 3))
    io.write_tum_trajectory(seq / "pred_traj.txt", np.arange(F), xyz, q)
    K = np.tile(np.array([[30.0, 0, W / 2], [0, 30.0, H / 2], [0, 0, 1]], dtype=np.float32), (F, 1, 1))
    np.savetxt(seq / "pred_intrinsics.txt", K.reshape(F, 9))
    c2w = [io.tumpose_to_c2w(np.concatenate([xyz[i], q[i]])) for i in range(F)]
    names = [f"frame_{i:04d}.png" for i in range(F)]
    io.write_colmap_cameras_text(seq / "sparse/0/cameras.txt", (W, H), K)
    io.write_colmap_images_text(seq / "sparse/0/images.txt", c2w, names)
    imgs = rng.integers(0, 256, size=(F, H, W, 3), dtype=np.uint8)
    for i in range(F):
        Image.fromarray(imgs[i]).save(seq / "images" / names[i])
        np.save(seq / "depth_maps" / f"frame_{i:04d}.npy", rng.uniform(1, 4, size=(H, W)).astype(np.float32))
        np.save(seq / "confidence_maps" / f"conf_{i:04d}.npy", rng.uniform(0, 3, size=(H, W)).astype(np.float32))
        np.save(seq / "dyna_avg" / f"dyna_avg_{i:04d}.npy", rng.uniform(0, 1, size=(H, W)).astype(np.float32))
    Image.fromarray((rng.uniform(size=(H, W)) > 0.5).astype(np.uint8) * 255).save(seq / "dynamic_masks" / "dynamic_mask_0000.png")
    s = io.load_sequence(str(seq))
    assert s["images"].shape == (F, 3, H, W) and s["depths"].shape == (F, H, W) and s["K"].shape == (F, 3, 3)
    assert np.array_equal((s["images"].numpy() * 255).round().astype(np.uint8), imgs.transpose(0, 3, 1, 2))
    assert np.allclose(s["cam2world"].numpy(), np.stack(c2w), atol=1e-6) and (s["W"], s["H"]) == (W, H)
    assert np.allclose(s["K"][:, 0, 2].numpy(), W / 2) and np.allclose(s["K"][:, 0, 0].numpy(), 30.0, rtol=1e-6)
    # w2c pose tensors: rebuilding the matrix from (quat, t) gives the inverse of the camera-to-world pose
    from das3r_amd.camera import camera_from_tensor
    for i in range(F):
        assert np.allclose(camera_from_tensor(s["w2c_pose7"][i]).numpy(), np.linalg.inv(c2w[i]), atol=1e-5)
    assert s["dynamic_masks"][0] is not None and s["dynamic_masks"][1] is None and s["names"] == names


def test_write_sequence_dir_round_trip(tmp_path):
    """write_sequence_dir -> load_sequence gives the sequence back (images to 8 bits, everything else exactly / to fp32 text precision)."""
    rng = np.random.default_rng(7)
    F, H, W = 4, 10, 16
    q = rng.normal(size=(F, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    c2w = np.stack([io.tumpose_to_c2w(np.concatenate([rng.normal(size=3), q[i]])) for i in range(F)]).astype(np.float32)
    K = np.tile(np.array([[25.0, 0, W / 2], [0, 27.0, H / 2], [0, 0, 1]], dtype=np.float32), (F, 1, 1))
    seq = dict(images=torch.from_numpy(rng.integers(0, 256, size=(F, 3, H, W)).astype(np.float32) / 255.0),
               depths=torch.from_numpy(rng.uniform(1, 4, size=(F, H, W)).astype(np.float32)),
               confs=torch.from_numpy(rng.uniform(0, 3, size=(F, H, W)).astype(np.float32)),
               dyna_avg=torch.from_numpy(rng.uniform(0, 1, size=(F, H, W)).astype(np.float32)),
               K=torch.from_numpy(K), cam2world=torch.from_numpy(c2w), W=W, H=H)
    names = io.write_sequence_dir(seq, str(tmp_path / "seq"))
    s = io.load_sequence(str(tmp_path / "seq"))
    assert s["names"] == names and (s["W"], s["H"]) == (W, H)
    assert np.array_equal((s["images"].numpy() * 255).round(), (seq["images"].numpy() * 255).round())
    for k in ("depths", "confs", "dyna_avg"):
        assert np.array_equal(s[k].numpy(), seq[k].numpy()), k
    assert np.allclose(s["cam2world"].numpy(), c2w, atol=2e-6)
    assert np.allclose(s["K"][:, 0, 0].numpy(), 25.0, rtol=1e-6) and np.allclose(s["K"][:, 1, 1].numpy(), 27.0, rtol=1e-6)


def test_save_poses_npy(tmp_path):
    poses = torch.tensor([[1.0, 0, 0, 0, 0.1, 0.2, 0.3], [0.9, 0.1, -0.2, 0.3, 1, 2, 3]])
    m = io.save_poses_npy(tmp_path / "pose" / "pose_4000.npy", poses)
    back = np.load(tmp_path / "pose" / "pose_4000.npy")
    assert back.shape == (2, 4, 4) and np.array_equal(back, m) and np.allclose(back[0][:3, 3], [0.1, 0.2, 0.3])
    assert np.allclose(back[1][:3, :3] @ back[1][:3, :3].T, np.eye(3), atol=1e-6)
