# oracle/ref_taps.mk -- build the only part of the reference that compiles without OpenCV:
# its file-static tap functions.  Sources are read where they lie under $(REF); the sliced
# lines and the binary go to oracle/_ref/ only (git-ignored).  Run:  make -f ref_taps.mk
REF ?= /root/reference
OUT := _ref

$(OUT)/ref_taps: ref_taps_driver.cpp $(REF)/cvsteer/SteerableFiltersG2.cpp $(REF)/cvsteer/SteerableFiltersG4.cpp
	mkdir -p $(OUT)
	sed -n '35,42p' $(REF)/cvsteer/SteerableFiltersG2.cpp > $(OUT)/g2_tap_lines.inc
	sed -n '34,45p' $(REF)/cvsteer/SteerableFiltersG4.cpp > $(OUT)/g4_tap_lines.inc
	g++ -std=c++11 -O0 -ffp-contract=off -I$(OUT) ref_taps_driver.cpp -o $@

golden: $(OUT)/ref_taps
	$(OUT)/ref_taps > ../tests/golden/taps_ref.json
