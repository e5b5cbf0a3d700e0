"""Page ingestion (yomitoku_amd/data): the reference's tests/test_data.py::test_load_image / test_load_pdf cases on files
generated here (the reference's fixture images are not in the repo), plus the staging ring on a HIP device."""
import numpy as np
import pytest


@pytest.fixture()
def files(tmp_path):
    from PIL import Image

    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(80, 120, 3), dtype=np.uint8)
    paths = {}
    for ext in ("jpg", "png", "tiff", "bmp"):
        paths[ext] = tmp_path / f"test.{ext}"
        Image.fromarray(rgb).save(paths[ext])
    paths["gray"] = tmp_path / "test_gray.jpg"
    Image.fromarray(rgb[:, :, 0]).save(paths["gray"])
    paths["rgba"] = tmp_path / "rgba.png"
    Image.fromarray(np.dstack([rgb, np.full((80, 120), 200, np.uint8)]), "RGBA").save(paths["rgba"])
    paths["multi"] = tmp_path / "sampldoc.tif"
    frames = [Image.fromarray(np.roll(rgb, k * 7, axis=1)) for k in range(3)]
    frames[0].save(paths["multi"], save_all=True, append_images=frames[1:])
    paths["small"] = tmp_path / "small.jpg"
    Image.fromarray(rgb[:20, :20]).save(paths["small"])
    paths["txt"] = tmp_path / "test.txt"
    paths["txt"].write_text("not an image")
    paths["invalid"] = tmp_path / "invalid.jpg"
    paths["invalid"].write_bytes(b"\\x00\\x01garbage")
    paths["pdf"] = tmp_path / "test.pdf"
    paths["pdf"].write_bytes(b"%PDF-1.4\\n%%EOF\\n")
    paths["rgb"] = rgb
    return paths


def test_load_image(files):
    from yomitoku_amd.data import load_image

    with pytest.raises(FileNotFoundError):
        load_image("dummy.jpg")
    for bad in ("txt", "small", "pdf", "invalid"):
        with pytest.raises(ValueError):
            load_image(str(files[bad]))
    for key in ("jpg", "png", "tiff", "bmp", "gray", "rgba", "multi"):
        pages = load_image(str(files[key]))
        assert len(pages) >= 1
        for page in pages:
            assert page.ndim == 3 and page.shape[2] == 3 and page.shape[0] > 32 and page.shape[1] > 32 and page.dtype == np.uint8
    assert len(load_image(str(files["multi"]))) == 3
    # lossless formats: the page is the image, channels reversed to BGR
    assert np.array_equal(load_image(str(files["png"]))[0], files["rgb"][:, :, ::-1])
    assert np.array_equal(load_image(str(files["multi"]))[1], np.roll(files["rgb"], 7, axis=1)[:, :, ::-1])


def test_load_pdf_contract(files):
    from yomitoku_amd.data import load_pdf

    with pytest.raises(FileNotFoundError):
        load_pdf("dummy.pdf")
    with pytest.raises(ValueError):
        load_pdf(str(files["txt"]))
    for key in ("jpg", "png", "tiff", "bmp", "gray"):
        with pytest.raises(ValueError):
            load_pdf(str(files[key]))
    try:
        import pypdfium2  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="pypdfium2"):  # rasterisation is blocked on the missing package, loudly
            load_pdf(str(files["pdf"]))
    else:
        with pytest.raises(ValueError):
            load_pdf(str(files["pdf"]))  # not a real PDF


@pytest.mark.gpu
def test_page_stager_and_stream(files, dev):
    import torch

    from yomitoku_amd.data import PageStager, load_image, stream_pages

    stager = PageStager(dev, slots=2)
    rng = np.random.default_rng(1)
    pages = [rng.integers(0, 256, size=(int(rng.integers(40, 900)), int(rng.integers(40, 700)), 3), dtype=np.uint8) for _ in range(9)]
    pages.append(pages[0][:, :, ::-1])  # a strided BGR view, as load_image returns
    on_dev = [stager.upload(p) for p in pages]  # ring of 2: slots are reused 5 times
    torch.cuda.synchronize()
    for host, d in zip(pages, on_dev):
        assert d.dtype == torch.uint8 and d.is_contiguous() and np.array_equal(d.cpu().numpy(), host)
    with pytest.raises(ValueError):
        stager.upload(np.zeros((10, 10), np.uint8))
    order = [str(files[k]) for k in ("png", "multi", "bmp")]
    got = list(stream_pages(order, dev, prefetch=2))
    assert [(p, k) for p, k, _, _ in got] == [(order[0], 0), (order[1], 0), (order[1], 1), (order[1], 2), (order[2], 0)]
    for path, k, host, d in got:
        assert np.array_equal(d.cpu().numpy(), host) and np.array_equal(host, load_image(path)[k])
    with pytest.raises(ValueError):
        list(stream_pages([str(files["small"])], dev))
