"""End-to-end known answers of the reference (README.md:244-251 of baudm/parseq: `./read.py pretrained=parseq --images
demo_images/*`).  They need the released PARSeq-S weights and the six demo images, neither of which is reachable from the
build container or the GPU box (no network; /root/reference does not travel), so the test is skipped unless both are
supplied:

    PARSEQ_WEIGHTS=/path/to/parseq-bb5792a6.pt PARSEQ_DEMO_IMAGES=/path/to/demo_images pytest tests/test_known_answers.py -m gpu

It is the one check that would pin the ViT encoder's restatement at the timm boundary by results rather than by structure
(DESIGN.md section 3)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

EXPECTED = {                      # README.md:246-251 of the reference
    'art-01107.jpg': 'CHEWBACCA',
    'coco-1166773.jpg': 'Chevrol',
    'cute-184.jpg': 'SALMON',
    'ic13_word_256.png': 'Verbandsteffe',
    'ic15_word_26.png': 'Kaopa',
    'uber-27491.jpg': '3rdAve',
}


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_demo_images_read_as_the_reference_readme_says(precision):
    weights, images = os.environ.get('PARSEQ_WEIGHTS'), os.environ.get('PARSEQ_DEMO_IMAGES')
    if not weights or not images or not os.path.exists(weights) or not os.path.isdir(images):
        pytest.skip('released weights / demo images not available (set PARSEQ_WEIGHTS and PARSEQ_DEMO_IMAGES)')
    from parseq_amd import create_model
    from read import read_files
    model = create_model('parseq', precision=precision)
    model.model.load_state_dict(torch.load(weights, map_location='cpu'))       # the released file is the inner model's state_dict
    model = model.eval().to('cuda')
    files = [os.path.join(images, name) for name in EXPECTED]
    got = {os.path.basename(f): label for f, label, _ in read_files(model, files)}
    assert got == EXPECTED
