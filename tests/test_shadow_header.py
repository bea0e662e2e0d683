"""The checker libraries under oracle/_ref are the reference's own .cpp files compiled beside a stand-in for its
Win32 platform header (oracle/shadow/LightTrack.h: the image has no <windows.h>, and the reference hard-wires a
600 x 600 screen).  This test keeps that stand-in honest mechanically: token for token it must equal lines 13-52 of
the reference's LightTrack.h up to the documented differences, and the HW_set_int that oracle/ref_driver.cpp
supplies must equal the reference's (LightTrack.cpp:82-87).  Runs only where the reference tree is present."""
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")


def _tokens(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+|\.\.\.|[^\sA-Za-z_0-9]", text)


def _lines(path, first, last):
    with open(path, errors="replace") as f:
        return "".join(f.readlines()[first - 1:last])


def test_shadow_header_equals_the_reference_lines_up_to_the_documented_differences():
    ref = _lines(os.path.join(REF, "LightTrack.h"), 13, 52)
    # documented differences, applied to the reference's text:
    #  1. the _16BPP_ branch is dropped (the checker build is _32BPP_ only and says so with #error)
    ref = re.sub(r"#if defined\(_16BPP_\).*?#endif", "", ref, flags=re.S)
    #  2. the screen size comes from the compiler command line instead of the two hard-wired defines
    ref = re.sub(r"#define HW_SCREEN_X_SIZE 600\s*\n#define HW_SCREEN_Y_SIZE 600[^\n]*\n", "", ref)
    #  3. `lng` is parenthesised in the two copy macros that multiply it
    ref = ref.replace("memcpy(dst,src,lng*sizeof(int))", "memcpy(dst,src,(lng)*sizeof(int))")
    ref = ref.replace("memcpy(dst,src,lng*sizeof(HW_16_bit))", "memcpy(dst,src,(lng)*sizeof(HW_16_bit))")
    shadow = open(os.path.join(ROOT, "oracle", "shadow", "LightTrack.h")).read()
    # the stand-in's own scaffolding: include guard, <string.h>, the two #error guards, the HW_error prototype
    # (LightTrack.h:68 in the reference, outside the compared lines)
    shadow = re.sub(r"#ifndef LT_SHADOW_LIGHTTRACK_H\s*\n#define LT_SHADOW_LIGHTTRACK_H\s*\n", "", shadow)
    shadow = shadow.replace("#include <string.h>", "")
    shadow = re.sub(r"#else\s*\n#error the checker build is _32BPP_ only\s*\n", "", shadow)
    shadow = re.sub(r"#ifndef HW_SCREEN_X_SIZE\s*\n#error[^\n]*\n#endif\s*\n", "", shadow)
    shadow = shadow.replace("void HW_error(char *s,...);", "")
    shadow = re.sub(r"#endif\s*$", "", shadow.rstrip() + "\n")
    assert _tokens(shadow) == _tokens(ref)
    # and the prototype that was set aside is the reference's own
    assert "void HW_error(char *s,...);" in _lines(os.path.join(REF, "LightTrack.h"), 68, 68)
    # the reference really hard-wires the size the stand-in replaces
    assert _tokens(_lines(os.path.join(REF, "LightTrack.h"), 44, 45)) == _tokens("#define HW_SCREEN_X_SIZE 600\n#define HW_SCREEN_Y_SIZE 600")


def _function(text, signature):
    i = text.index(signature)
    j = text.index("}", i)
    return text[i:j + 1]


def test_hw_set_int_of_the_driver_equals_the_reference():
    ref = _function(_lines(os.path.join(REF, "LightTrack.cpp"), 82, 87), "void HW_set_int(")
    drv = _function(open(os.path.join(ROOT, "oracle", "ref_driver.cpp")).read(), "void HW_set_int(")
    assert _tokens(drv) == _tokens(ref)


def test_build_recipe_compiles_the_reference_files_where_they_lie():
    """every source on the path is linked from /root/reference (no copy in the repository), with the configuration's defines"""
    recipe = open(os.path.join(ROOT, "oracle", "build_ref.sh")).read()
    for d in ("-D_RGB_", "-D_32BPP_", "-D_FIXED_", "-D_Z_BUFFER_", "-D_MONO_", "-fwrapv"):
        assert d in recipe
    for f in ("Clipp2D", "Clipp3D", "Colour", "EngBase", "EngPoly", "GraphicsBase", "GraphicsPoly", "TransLn"):
        assert f in recipe
        assert not os.path.exists(os.path.join(ROOT, "oracle", f + ".cpp"))
